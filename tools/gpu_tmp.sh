timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids" | tail -3
python bench.py --config config/cfg_kitti_fm_joint_inpaint_disentangle_distill_full_colorize.py --steps 3 --warmup 2 --miopen-find off --no-cpu-baseline --no-roofline 2>&1 | grep -v "amdgpu.ids\|Warning\|run_backward" | tail -1 | cut -c1-420
