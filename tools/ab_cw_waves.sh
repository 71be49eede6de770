# Waves per SIMD of the chain-per-wavefront cluster kernel (kernel experiments; profiles/r04/experiments/ab_cw_waves.txt):
#   bash tools/build_variant.sh w5 pstat_cluster_cw.hip pstat_cluster_cw.o -ffp-contract=fast -mllvm -disable-machine-licm -DPSTAT_CW_WAVES=5
set -e
B=polymer_stats_amd/csrc/build
python tests/first_divergence_cluster.py wave | tail -5
for lib in default var_w5; do
  echo "== $lib"
  if [ $lib = default ]; then python tools/time_cluster_cw.py 20000 100 1,4,16 wave
  else PSTAT_LIB=$B/$lib/libpstat.so python tools/time_cluster_cw.py 20000 100 1,4,16 wave; fi
done
