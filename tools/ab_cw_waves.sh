# Waves per SIMD of the chain-per-wavefront cluster kernel (kernel experiments; profiles/r04/experiments/ab_cw_waves.txt):
#   for w in 3 4; do bash tools/build_variant.sh w$w pstat_cluster_cw.hip pstat_cluster_cw.o -ffp-contract=fast -DPSTAT_CW_WAVES=$w; done
set -e
B=polymer_stats_amd/csrc/build
python tests/first_divergence_cluster.py wave | tail -5
for lib in default var_w3 var_w4; do
  echo "== $lib"
  if [ $lib = default ]; then python tools/time_cluster_cw.py 20000 100 1,4,16 wave
  else PSTAT_LIB=$B/$lib/libpstat.so python tools/time_cluster_cw.py 20000 100 1,4,16 wave; fi
done
