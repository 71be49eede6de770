#!/bin/bash
# Three of the reference's sweep drivers (the phase scan at two ensemble sizes) end to end through tools/run_sweep.py (one GPU), timed:
#   bash tools/reference_sweeps.sh <outdir>      -> <outdir>/*.log, <outdir>/*.csv (aggregated like scripts/aggregate_mcmc.jl)
set -euo pipefail
out=${1:-gpurun_out/sweeps}
mkdir -p "$out"
w=$(mktemp -d /tmp/sweeps.XXXXXX)
t() { local tag=$1; shift; local s=$(date +%s%N); "$@" 2> "$out/$tag.log"; local ms=$(( ($(date +%s%N) - s) / 1000000 )); echo "$tag: $ms ms wall (process start to exit), $(ls "$w/$tag" | grep -c '\.out$') .out files; $(grep '^# rank 0 of' "$out/$tag.log")" | tee -a "$out/summary.txt"; }
: > "$out/summary.txt"
# run/interacting_dielectric_study.jl:19-43 -- 5 760 cases, all-pairs energy, n = 100 and 200, 500 000 steps each, one chain per case
t interacting_dielectric_study python tools/run_sweep.py "$w/interacting_dielectric_study" --main mcmc_eap_chain --num-chains 1 --seed 1 \
  --axis b=0.5,1,2 --axis n=100,200 --axis Fx=0,0.5,1,2 --axis Fz=0,0.5,1,2 --axis kT=1 --axis E0=0.1,1,10 \
  --axis K1=0,0.1,0.5,1,2 --axis K2=0,0.1,0.5,1,2 --skip 'K1==K2' --name E0,K1,K2,kT,Fz,Fx,n,b \
  --aggregate "$out/interacting_dielectric_study.csv" \
  -- --chain-type dielectric --energy-type interacting --num-steps 500000 -v 2
# run/K1_E0-kT-phase.jl:17-45 -- 546 grid points x 5 runs, clustering main, Ising, 5 x 100 000 burn-in + 2 500 000 steps; 16 chains per case
t K1_E0-kT-phase python tools/run_sweep.py "$w/K1_E0-kT-phase" --main mcmc_clustering_eap_chain --num-chains 16 --seed 2 \
  --axis b=1 --axis n=100 --axis Fx=0 --axis Fz=0 --axis kT='10^(-2:0.2:2)' --axis E0=0:0.2:5 --axis K1=1 --axis K2=0 --axis kappa=0 \
  --axis run=1:5 --name E0,K1,K2,kT,Fz,Fx,n,b,kappa,run:raw --aggregate "$out/K1_E0-kT-phase.csv" --aggregate-args '*.out,dielectric,true,true' \
  -- --chain-type dielectric --energy-type Ising --num-steps 2500000 --burn-in 100000 -v 2 --stepout 250
# the same sweep as the reference launches it: ONE chain per case (546 x 5 independent runs): one chain per wavefront (pstat_cluster_cw.hip)
t K1_E0-kT-phase_1chain python tools/run_sweep.py "$w/K1_E0-kT-phase_1chain" --main mcmc_clustering_eap_chain --num-chains 1 --seed 2 \
  --axis b=1 --axis n=100 --axis Fx=0 --axis Fz=0 --axis kT='10^(-2:0.2:2)' --axis E0=0:0.2:5 --axis K1=1 --axis K2=0 --axis kappa=0 \
  --axis run=1:5 --name E0,K1,K2,kT,Fz,Fx,n,b,kappa,run:raw --aggregate "$out/K1_E0-kT-phase_1chain.csv" --aggregate-args '*.out,dielectric,true,true' \
  -- --chain-type dielectric --energy-type Ising --num-steps 2500000 --burn-in 100000 -v 2 --stepout 250
# run/noninteracting-compare-with-clustering_2021-09-24.jl:19-33 -- 29 forces x 6 fields (BASELINE configs[1]'s grid), 1 000 000 steps, 64 chains per point
t noninteracting_force_sweep python tools/run_sweep.py "$w/noninteracting_force_sweep" --main mcmc_eap_chain --num-chains 64 --seed 3 \
  --axis Fz='0:0.05:1,1.5:0.5:5' --axis kT=1 --axis E0=0:1:5 --axis K1=1 --axis K2=0 --axis Fx=0 --axis n=100 --axis b=1 --name E0,K1,K2,kT,Fz,Fx,n,b \
  --aggregate "$out/noninteracting_force_sweep.csv" \
  -- --chain-type dielectric --energy-type noninteracting --num-steps 1000000 -v 2
rm -rf "$w"
cat "$out/summary.txt"
