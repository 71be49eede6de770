# A/B of the f64 sweep's step (kernel experiments): the library as built against a variant holding the previous pstat_sweep_f64g.o
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
