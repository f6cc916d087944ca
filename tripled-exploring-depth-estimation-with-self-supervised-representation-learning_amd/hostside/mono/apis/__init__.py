from .env import get_root_logger, init_dist, set_random_seed  # noqa: F401
from .trainer import batch_processor, build_optimizer, change_input_variable, train_mono  # noqa: F401
