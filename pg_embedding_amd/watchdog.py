"""A hard time limit for anything that launches kernels from a script: experiment drivers, benchmarks, stress loops.

    from pg_embedding_amd import watchdog
    watchdog.arm()            # reads and removes `--timeout SECONDS` from sys.argv (default 900)

Before the first launch it (1) turns on the library's own kernel watchdog (HNSW_GPU_WATCHDOG_S: a search launch that runs
longer than that is asked to end through its workspace's abort word, include/hnsw_gpu.h), and (2) starts a thread that, when
the limit is reached, dumps every Python thread's stack, asks every launch in flight to end (hnsw_gpu_abort_all), gives the
device two seconds to drain and exits the process with status 124.  A device run that hangs costs one case, not the round:
nothing here relies on the main thread coming back from a blocked HIP call."""
from __future__ import annotations

import faulthandler
import os
import sys
import threading
import time

_armed = False


def _expire(seconds: float) -> None:
    time.sleep(seconds)
    sys.stderr.write(f"\n[watchdog] {seconds:.0f} s limit reached: stacks follow, asking the launches in flight to end\n")
    sys.stderr.flush()
    try:
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
    except Exception:
        pass
    try:
        from . import _lib
        if _lib._gpu is not None:
            n = _lib._gpu.hnsw_gpu_abort_all()
            sys.stderr.write(f"[watchdog] abort word set on {n} search workspace(s)\n")
    except Exception as e:                                   # the exit below must happen whatever this did
        sys.stderr.write(f"[watchdog] could not reach the library: {e}\n")
    sys.stderr.flush()
    time.sleep(2.0)
    os._exit(124)


def arm(default_seconds: float = 900.0, kernel_seconds: float = 120.0, env_sync: bool = True) -> float:
    """Start the limit (once per process).  Returns the limit in seconds.
    env_sync: experiment scripts flip HNSW_GPU_* knobs through os.environ while they run; the library reads its environment once,
    so pg_embedding_amd forwards such changes before every launch (PGEMB_ENV_SYNC, _lib.sync_env).  bench.py switches it off."""
    global _armed
    if env_sync:
        os.environ.setdefault("PGEMB_ENV_SYNC", "1")
    seconds = default_seconds
    if "--timeout" in sys.argv:
        i = sys.argv.index("--timeout")
        seconds = float(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    os.environ.setdefault("HNSW_GPU_WATCHDOG_S", str(int(max(1, min(kernel_seconds, seconds)))))
    if not _armed:
        _armed = True
        threading.Thread(target=_expire, args=(seconds,), daemon=True, name="pgemb-watchdog").start()
    return seconds
