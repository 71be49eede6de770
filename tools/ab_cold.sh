# Ring depth of the in-memory clustering kernel on aligned (cold) chains (kernel experiments; profiles/r04/experiments/ab_cold.txt):
#   for v in "d4 -DPSTAT_GM_D=4 -DPSTAT_GM_CAPT=10" "d6 -DPSTAT_GM_D=6 -DPSTAT_GM_CAPT=7" "prof -DPSTAT_GM_PROF"; do set -- $v; t=$1; shift;
#     bash tools/build_variant.sh $t pstat_cluster_gm.hip pstat_cluster_gm.o -ffp-contract=fast "$@"; done
set -e
B=polymer_stats_amd/csrc/build
for cfg in "E0=3 KT=0.1" "E0=1 KT=1"; do
  for lib in default var_d4 var_d6; do
    for chains in 65536 4096; do
      if [ $lib = default ]; then env $cfg K1=1 K2=0 python tools/time_cluster.py ising f64 3000 100 $chains
      else env $cfg K1=1 K2=0 PSTAT_LIB=$B/$lib/libpstat.so python tools/time_cluster.py ising f64 3000 100 $chains; fi
    done
  done
done
E0=3 KT=0.1 K1=1 K2=0 PSTAT_LIB=$B/var_prof/libpstat.so python tools/time_cluster.py ising f64 3000 100 4096 | tail -3
