# Evidence for the chain-per-wavefront kernel of the clustering main at HEAD (GPU box) -> gpurun_out/evidence_cw/
set -uo pipefail
o=gpurun_out/evidence_cw; mkdir -p $o
python -m pytest tests -q -m gpu > $o/gpu_tests.log 2>&1; tail -1 $o/gpu_tests.log
python -m pytest tests -q -m perf > $o/perf_tests.log 2>&1; tail -1 $o/perf_tests.log
python tests/fuzz_cluster_wave.py 600 2 > $o/fuzz_cluster_wave.txt 2>&1; tail -1 $o/fuzz_cluster_wave.txt
python tests/soak_cluster.py 40000 12 > $o/soak_cluster_wave.txt 2>&1; tail -4 $o/soak_cluster_wave.txt
{ echo "# n = 100"; python tools/time_cluster_cw.py 20000 100 1,2,4,8,16,32,64; echo "# n = 200"; python tools/time_cluster_cw.py 10000 200 1,4,16; } > $o/time_cluster_cw.txt 2>&1; cat $o/time_cluster_cw.txt
{ bash tools/pmc_cw.sh whole 1 20000; bash tools/pmc_cw.sh cold 1 20000; bash tools/pmc_cw.sh hot 1 20000; } > $o/pmc_cluster_wave.txt 2>&1; grep "grid\|INSTS_VALU \|INSTS_SALU\|INSTS_LDS\|WAVE_CYCLES\|ACTIVE_INST_ANY\|WAIT_ANY\|F64" $o/pmc_cluster_wave.txt
