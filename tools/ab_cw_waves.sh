# Waves per SIMD of the chain-per-wavefront cluster kernel, and what machine LICM costs it (kernel experiments;
# profiles/r04/experiments/ab_cw_waves.txt).  Variants (the default build asks for 4 waves with machine LICM off):
#   for w in 2 3 5; do bash tools/build_variant.sh w$w pstat_cluster_cw.hip pstat_cluster_cw.o -ffp-contract=fast -mllvm -disable-machine-licm -DPSTAT_CW_WAVES=$w; done
#   bash tools/build_variant.sh w4licm pstat_cluster_cw.hip pstat_cluster_cw.o -ffp-contract=fast -DPSTAT_CW_WAVES=4
set -e
B=polymer_stats_amd/csrc/build
for lib in var_w2 var_w3 default var_w5 var_w4licm; do
  echo "== $lib"
  if [ $lib = default ]; then python tools/time_cluster_cw.py 20000 100 1,16 wave
  else PSTAT_LIB=$B/$lib/libpstat.so python tools/time_cluster_cw.py 20000 100 1,16 wave; fi
done
