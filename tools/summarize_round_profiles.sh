#!/bin/bash
# gpurun_out/prof_<tag>_* (tools/collect_round_profiles.sh) -> profiles/<tag>/ and profiles/pmc_traffic.json
set -euo pipefail
tag=${1:-r04}
S="python tools/summarize_pmc.py"
$S gpurun_out/prof_${tag}_bench_f64 profiles/$tag bench_f64 --kernel sweep_kernel --updates 6553600000 --record f64_n100_c65536_s100000 > /dev/null
$S gpurun_out/prof_${tag}_bench_f32 profiles/$tag bench_f32 --kernel sweep_kernel --updates 6553600000 --record f32_n100_c65536_s100000 > /dev/null
$S gpurun_out/prof_${tag}_cluster_f64_ni profiles/$tag cluster_f64_ni --kernel cluster --updates 327680000 > /dev/null
$S gpurun_out/prof_${tag}_cluster_f64_ising profiles/$tag cluster_f64_ising --kernel cluster --updates 327680000 > /dev/null
for w in cluster_f32_ni cluster_f32_mem_n200; do
  if [ -d gpurun_out/prof_${tag}_$w ]; then $S gpurun_out/prof_${tag}_$w profiles/$tag $w --kernel cluster --updates 327680000 > /dev/null; fi
done
$S gpurun_out/prof_${tag}_cfg_C1 profiles/$tag cfg_C1_sweep_f64_n20 --kernel sweep_kernel --updates 6553600000 --record cfg_C1 > /dev/null
$S gpurun_out/prof_${tag}_cfg_C3 profiles/$tag cfg_C3_sweep_f64_polar_n100 --kernel sweep_kernel --updates 6553600000 --record cfg_C3 > /dev/null
$S gpurun_out/prof_${tag}_cfg_C4 profiles/$tag cfg_C4_interacting_f64_n64 --kernel interacting_kernel --updates 327680000 --record cfg_C4 > /dev/null
$S gpurun_out/prof_${tag}_cfg_C5 profiles/$tag cfg_C5_sweep_f64_ising_n200_grid --kernel sweep_kernel --updates 1397760000 --record cfg_C5 > /dev/null
$S gpurun_out/prof_${tag}_cluster_wave_f64_n100 profiles/$tag cluster_wave_f64_n100 --kernel cluster_wave --updates 16384000 > /dev/null
if [ -d gpurun_out/prof_${tag}_cluster_cw_phase ]; then
  $S gpurun_out/prof_${tag}_cluster_cw_phase profiles/$tag cluster_cw_phase_scan_n100 --kernel cluster_cw --updates 54600000 > /dev/null
fi

if [ -d gpurun_out/prof_${tag}_sweep_f64_ni_n200 ]; then
  $S gpurun_out/prof_${tag}_sweep_f64_ni_n200 profiles/$tag sweep_f64_ni_n200 --kernel sweep_kernel --updates 3276800000 > /dev/null
fi
if [ -f gpurun_out/calib/stdout.txt ]; then python tools/summarize_calib.py gpurun_out/calib profiles/$tag/fetch_calib.txt > /dev/null; fi
ls profiles/$tag
