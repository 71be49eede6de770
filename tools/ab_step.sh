# A/B of the f64 sweep's step (kernel experiments; profiles/r04/experiments/ab_step.txt): the library as built against a variant
# holding a pstat_sweep_f64g.o compiled from an earlier commit's csrc/ (git archive <commit> polymer_stats_amd/csrc include | tar -x -C /tmp/old;
# hipcc ... -DPSTAT_PART=4 -c -o polymer_stats_amd/csrc/build/var_old/pstat_sweep_f64g.o /tmp/old/polymer_stats_amd/csrc/pstat_kernels.hip; link with the other objects)
set -e
V=polymer_stats_amd/csrc/build/var_old/libpstat.so
for i in 1 2 3; do
  python tools/time_sweep.py f64 100 65536 100000 5
  PSTAT_LIB=$V python tools/time_sweep.py f64 100 65536 100000 5
done
python tools/time_sweep.py f64 200 65536 20000 5 2
PSTAT_LIB=$V python tools/time_sweep.py f64 200 65536 20000 5 2
python tools/time_sweep.py f64 200 65536 50000 5 0
PSTAT_LIB=$V python tools/time_sweep.py f64 200 65536 50000 5 0
