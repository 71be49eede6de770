# The round's closing evidence at HEAD (GPU box) -> gpurun_out/evidence_final/: the bench line twice, the whole GPU suite, the perf
# tripwires, and the long forms of the trajectory fuzzes and soaks of every f64 kernel family.
set -uo pipefail
o=gpurun_out/evidence_final; mkdir -p $o
python bench.py > $o/bench_line.json 2> $o/bench_stderr.txt; tail -c 600 $o/bench_line.json; echo
python -m pytest tests -q -m gpu > $o/gpu_tests.log 2>&1; tail -1 $o/gpu_tests.log
python -m pytest tests -q -m perf > $o/perf_tests.log 2>&1; tail -1 $o/perf_tests.log
python tests/fuzz_f64.py 800 > $o/fuzz_f64.txt 2>&1; tail -1 $o/fuzz_f64.txt
python tests/fuzz_packed.py 3000 1 > $o/fuzz_packed.txt 2>&1; tail -1 $o/fuzz_packed.txt
python tests/fuzz_cluster_wave.py 600 2 > $o/fuzz_cluster_wave.txt 2>&1; tail -1 $o/fuzz_cluster_wave.txt
PSTAT_F64_STATE=global python tests/soak_cluster.py 40000 12 > $o/soak_cluster.txt 2>&1; tail -4 $o/soak_cluster.txt
python tests/soak_cluster.py 40000 12 > $o/soak_cluster_wave.txt 2>&1; tail -4 $o/soak_cluster_wave.txt
python bench.py > $o/bench_line_2.json 2>> $o/bench_stderr.txt; python -c "
import json,sys
for f in ('$o/bench_line.json','$o/bench_line_2.json'):
    d=json.loads(open(f).read().strip()); print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['phase_scan']['us_per_step'], [round(c['value']/1e9,2) for c in d['configs']])
"
