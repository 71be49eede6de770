set -e
V=polymer_stats_amd/csrc/build/var_narrow/libpstat.so
for i in 1 2; do
  python tools/time_sweep.py f64 100 65536 100000 5
  PSTAT_LIB=$V python tools/time_sweep.py f64 100 65536 100000 5
  UNIFORM_BITS=23 python tools/time_sweep.py f64 100 65536 100000 5
done
