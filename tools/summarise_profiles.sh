#!/bin/bash
# gpurun_out/prof/{pm_fast,pm_exact,pm_cli,ps_fast,ps_exact} (tools/final_cycle.sh) -> profiles/rNN_*.txt and profiles/traffic.json
# usage: tools/summarise_profiles.sh r04
R=${1:-r04}
export PM_ITERS=8 PM_SAMPLES=8
python tools/profile_summary.py pm_fast  ${R}_pm_step_fast  "config 3, fast arithmetic: 16 views 1920x1080, 8 x (2+8), 7x7, groups of 4 views per launch" "pm_step_fast_kernel<7,4>" 4x1920x1080 > profiles/${R}_pm_step_fast.txt
python tools/profile_summary.py pm_exact ${R}_pm_step_exact "config 3, exact arithmetic (the classes' default): 16 views 1920x1080, 8 x (2+8), 7x7, whole batch per launch" "pm_step_kernel<7,4>" 16x1920x1080 > profiles/${R}_pm_step_exact.txt
PM_ITERS=3 python tools/profile_summary.py pm_cli ${R}_pm_step_cli_defaults "run_reconstruction.py defaults: exact arithmetic, 11x11, 3 x (2+8), 16 views 1008x756 (4032x3024 at scale 0.25)" "pm_step_kernel<11,4>" 16x1008x756 > profiles/${R}_pm_step_cli_defaults.txt
python tools/profile_summary.py ps_fast  ${R}_plane_sweep_fast  "config 2, fast arithmetic: 8 views 1280x720, 64 planes, 5x5, 6 neighbours" "plane_sweep_fast_kernel<5,6>" 8x1280x720 > profiles/${R}_plane_sweep_fast.txt
python tools/profile_summary.py ps_exact ${R}_plane_sweep_exact "config 2, exact arithmetic (the stereo class's default)" "plane_sweep_kernel<5,6>" 8x1280x720 > profiles/${R}_plane_sweep_exact.txt
