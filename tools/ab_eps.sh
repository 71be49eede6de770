# A/B of the 53-bit Metropolis eps on the f64 headline kernel (kernel experiments; profiles/r04/experiments/ab_eps.txt):
# the default library against a variant whose pstat_sweep_f64g.o is built with -DPSTAT_NARROW_EPS (the 23-bit-only filter):
#   bash tools/build_variant.sh narrow pstat_kernels.hip pstat_sweep_f64g.o -ffp-contract=fast -DPSTAT_PART=4 -DPSTAT_NARROW_EPS
set -e
V=polymer_stats_amd/csrc/build/var_narrow/libpstat.so
for i in 1 2; do
  python tools/time_sweep.py f64 100 65536 100000 5
  PSTAT_LIB=$V python tools/time_sweep.py f64 100 65536 100000 5
  UNIFORM_BITS=23 python tools/time_sweep.py f64 100 65536 100000 5
done
