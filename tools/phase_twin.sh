#!/bin/bash
# run/K1_E0-kT-phase.jl:17-45 through tools/run_sweep.py (546 grid points x 5 runs, clustering main, Ising; 16 chains per case),
# timed; STEPS / BURN scale the run (reference: 2 500 000 / 100 000).  PSTAT_PACK=0 keeps every workgroup inside one case.
#   bash tools/phase_twin.sh <tag> [steps] [burn]
set -euo pipefail
tag=$1; steps=${2:-2500000}; burn=${3:-100000}
w=$(mktemp -d /tmp/twin.XXXXXX)
s=$(date +%s%N)
python tools/run_sweep.py "$w" --main mcmc_clustering_eap_chain --num-chains ${CHAINS:-16} --seed 2 \
  --axis b=1 --axis n=100 --axis Fx=0 --axis Fz=0 --axis kT='10^(-2:0.2:2)' --axis E0=0:0.2:5 --axis K1=1 --axis K2=0 --axis kappa=0 \
  --axis run=1:5 --name E0,K1,K2,kT,Fz,Fx,n,b,kappa,run:raw \
  -- --chain-type dielectric --energy-type Ising --num-steps $steps --burn-in $burn -v 2 --stepout 250 2> "$w/log"
ms=$(( ($(date +%s%N) - s) / 1000000 ))
echo "$tag: steps=$steps burn=$burn PSTAT_PACK=${PSTAT_PACK:-auto}: $ms ms wall; $(grep '^# rank 0:' "$w/log")"
rm -rf "$w"
