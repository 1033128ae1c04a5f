#!/bin/bash
# GPU cycle 11: round-2 profiles (full config-3 schedule in both arithmetic modes, config 2)
PROF_TAG=pm_fast bash tools/profile.sh
PROF_TAG=pm_exact BENCH_ARGS="--no-planesweep --mode exact" PMC_GROUPS="pmc_fetch pmc_write" bash tools/profile.sh
PROF_TAG=ps_fast BENCH_ARGS="--workload planesweep" PMC_GROUPS="pmc_sq pmc_fetch pmc_write" bash tools/profile.sh
du -sh gpurun_out/prof
echo cycle-done
