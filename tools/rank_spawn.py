"""Start N rank processes of a script on ONE node and supervise them (bench.py --gpus N, tools/phase_scan.py --gpus N).

The parent makes no GPU call and never imports torch: the children are fresh processes started BEFORE anything touches
the GPU (on this pool a process that has initialised the GPU must not exec another program).  The reference's analogue
is `pmap` over worker processes (run/interacting_dielectric_study.jl:37-47, run/K1_E0-kT-phase.jl:19-45).

  * every rank's stderr is the parent's stderr; rank 0's stdout is captured (its JSON / CSV result), the other ranks'
    stdout goes to stderr, so no diagnostic is lost;
  * all ranks are polled together: as soon as one exits non-zero the others are terminated (then killed), so a rank that
    dies before the rendezvous cannot leave the rest sitting in init_process_group / a barrier with GPUs held;
  * the same clean-up runs on KeyboardInterrupt, on a parent-side exception and on the overall deadline
    (PSTAT_SPAWN_TIMEOUT seconds, default none)."""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(script: str, argv, nranks: int, what: str = "rank"):
    """Runs `python script argv...` as ranks 0..nranks-1.  Returns (rc, rank-0 stdout as str)."""
    port = free_port()
    deadline = float(os.environ.get("PSTAT_SPAWN_TIMEOUT", "0")) or None
    procs, out0 = [], []
    reader = None
    t0 = time.time()
    try:
        for r in range(nranks):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
        reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        failed = None
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                failed = f"{what} {bad[0][0]} exited with code {bad[0][1]}"
                break
            if all(rc == 0 for rc in rcs):
                break
            if deadline and time.time() - t0 > deadline:
                failed = f"deadline of {deadline:.0f} s passed"
                break
            time.sleep(0.05)
        if failed:
            print(f"{os.path.basename(script)}: {failed}; stopping the other ranks "
                  f"(exit codes so far: {[p.poll() for p in procs]})", file=sys.stderr, flush=True)
            return 1, ""
        reader.join(timeout=10)
        return 0, (out0[0].decode() if out0 and out0[0] else "")
    finally:
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t1 = time.time()
        for p in live:
            try:
                p.wait(timeout=max(0.1, 5 - (time.time() - t1)))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
