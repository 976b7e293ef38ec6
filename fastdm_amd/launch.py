"""One process per GPU without a launcher: start the ranks, watch ALL of them, fail fast.

`examples/evaluation/clip_score.py:346-365` of the reference starts one worker per GPU with torch.multiprocessing and joins them; here
the ranks are fresh child processes (so the parent never initialises HIP and nothing is ever re-exec'ed after a GPU call) and the
parent supervises them:

* every child is polled, not only rank 0 -- a rank that dies before or at rendezvous would otherwise leave its siblings blocked in
  `init_process_group` / the first collective until the RCCL timeout (minutes), and the caller with no output;
* on the first non-zero exit (or the overall timeout) the remaining ranks are terminated, then killed, and the call returns non-zero;
* rank 0's stdout is relayed (the benchmark's ONE JSON line); the other ranks' stdout goes to stderr instead of being dropped.

No torch import: the supervisor must stay a plain Python process."""
import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, local_rank: Optional[int] = None, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank if local_rank is None else local_rank), WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL / CUDA-tensor sharing across processes needs it on this driver
    return env


def supervise(cmds: Sequence[Sequence[str]], envs: Sequence[Dict[str, str]], timeout: Optional[float] = None, poll: float = 0.2,
              grace: float = 5.0, relay_rank0: bool = True) -> int:
    """Run one child per (cmd, env); returns 0 iff every child exited 0.  Rank 0's stdout is written to this process's stdout
    (relay_rank0), every other rank's stdout to stderr."""
    procs: List[subprocess.Popen] = []
    pumps: List[threading.Thread] = []
    out0: List[str] = []

    def pump(stream, rank):
        for line in stream:
            if rank == 0 and relay_rank0:
                out0.append(line)
            else:
                sys.stderr.write(f"[rank {rank}] {line}")
        stream.close()

    for r, (cmd, env) in enumerate(zip(cmds, envs)):
        p = subprocess.Popen(list(cmd), env=env, stdout=subprocess.PIPE, stderr=None, text=True)
        procs.append(p)
        t = threading.Thread(target=pump, args=(p.stdout, r), daemon=True)
        t.start()
        pumps.append(t)
    t0 = time.time()
    failed: List = []
    try:
        while True:
            rcs = [p.poll() for p in procs]
            failed = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if failed or all(rc == 0 for rc in rcs):
                break
            if timeout is not None and time.time() - t0 > timeout:
                failed = [("timeout", timeout)]
                break
            time.sleep(poll)
    finally:
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        t1 = time.time()
        while any(p.poll() is None for p in alive) and time.time() - t1 < grace:
            time.sleep(0.05)
        for p in alive:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=grace)
            except subprocess.TimeoutExpired:
                pass
    for t in pumps:
        t.join(timeout=grace)
    if out0:
        sys.stdout.write("".join(out0))
        sys.stdout.flush()
    if failed:
        rcs = [p.poll() for p in procs]
        print(f"launch: ranks failed (rank, exit code): {failed}; exit codes of all ranks after shutdown: {rcs}", file=sys.stderr)
        return 1
    return 0


def spawn_self(script: str, argv: Sequence[str], world: int, timeout: Optional[float] = None, same_device: bool = False) -> int:
    """Start `world` copies of `python script argv...`, rank r on GPU r (or all on GPU 0: `same_device`)."""
    port = free_port()
    cmds = [[sys.executable, os.path.abspath(script)] + list(argv) for _ in range(world)]
    envs = [rank_env(r, world, port, local_rank=0 if same_device else r) for r in range(world)]
    return supervise(cmds, envs, timeout=timeout)
