"""Shared pieces of the shipped training configs (mmcv python-config format; the model/data
keys are the reference's, see config/*.py in the reference).  Not a config itself."""


def schedule(lr, steps, total_epochs):
    return dict(
        total_epochs=total_epochs,
        learning_rate=lr,
        optimizer=dict(type="Adam", lr=lr, weight_decay=0),
        optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)),
        lr_config=dict(policy="step", warmup="linear", warmup_iters=500, warmup_ratio=1.0 / 3, step=steps, gamma=0.5),
    )


def runtime(tensorboard=False):
    hooks = [dict(type="TextLoggerHook")]
    if tensorboard:
        hooks.append(dict(type="TensorboardLoggerHook"))
    return dict(
        resume_from=None, finetune=None, load_from=None, validate=True, validate_interval=1,
        find_unused_parameters=True, checkpoint_config=dict(interval=1),
        log_config=dict(interval=50, hooks=hooks), dist_params=dict(backend="nccl"), log_level="INFO",
        workflow=[("train", 1)], syncbn=True, cudnn_benchmark=True,
    )
