"""TEST INFRASTRUCTURE: one rank of bench.py's multi-rank program (benchmarks/multirank.run_multi: ONE bundle split over the
ranks, per-step image-plane exchange, the start-up probe that chooses between the collective and the direct peer writes)
with engine.py on the HOST build of libprt and gloo between the ranks -- the Python of the N > 1 launch on a box without a
GPU.  Started by tests/test_hostemu_campaigns.py, one process per rank (RANK / WORLD_SIZE / MASTER_PORT in the
environment); prints one line `RANK r <json>`."""
import contextlib
import importlib.util
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch                                # noqa: E402
import torch.distributed as dist            # noqa: E402
from hostemu.engine_on_host import engine_on_host, _Stream       # noqa: E402


class Clock(object):
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def synchronize(self):
        pass

    def wait(self, stream=None):
        pass

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class Stream(_Stream):
    def __init__(self, device=None, priority=0):
        pass


def main():
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    with engine_on_host():
        torch.cuda.Event = Clock
        torch.cuda.Stream = Stream
        torch.cuda.stream = lambda s: contextlib.nullcontext()
        torch.cuda.set_stream = lambda s: None
        sys.argv = ["bench.py", "--gpus", str(world), "--backend", "gloo", "--rays-total", "4000", "--steps", "2", "--warmup", "1",
                    "--no-cpu-baseline"] + sys.argv[1:]
        spec = importlib.util.spec_from_file_location("bench_rank", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        args = bench.parse_args()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from benchmarks import multirank
        wd = types.SimpleNamespace(stage="", done=lambda: None)
        (line, _) = multirank.run_multi(args, torch.device("cpu"), world, rank, rank, wd)
        out = None
        if line is not None:
            out = {"config": {k: line["config"].get(k) for k in ("exchange", "exchange_probe_ms", "rays_total", "ms_gather",
                                                                  "ms_trace", "ms_total")},
                   "verified": line["verified"], "value": line["value"], "n_gpus": line["n_gpus"], "bytes": len(json.dumps(line))}
        print("RANK %d %s" % (rank, json.dumps(out)), flush=True)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
