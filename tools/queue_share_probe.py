#!/usr/bin/env python3
"""Does the audio tail really run beside the next step's front end?  HIP maps streams onto a few hardware queues; two
streams on one queue run one after the other.  The -M wbfm step (1024 streams x 16 x 262144 B) with 0..7 other streams
created in the process before the handle, and with the tail's stream at the default / lowest / highest priority."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    hip = C.CDLL("libamdhip64.so")
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    sys.argv = [sys.argv[0], "--workload", "wbfm"]
    a = bench.workload_args(bench.parse(), "wbfm")
    made = []
    for pre in (0, 1, 2, 3, 5, 8):
        while len(made) < pre:
            q = C.c_void_p()
            assert hip.hipStreamCreateWithFlags(C.byref(q), 1) == 0
            made.append(q)
        for prio in (0, -1, 1):
            j = bench.FmJob(a, dev, 0, 0)
            if prio:
                j.g.set_option("tail_priority", prio)
            launch_ms, step_ms = bench._timed(j, 60, 200)
            print(f"streams created before the handle {pre}  tail_priority {prio:2d}  front end {launch_ms:.4f} ms  step {step_ms:.4f} ms", flush=True)
            j.close()
            del j
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
