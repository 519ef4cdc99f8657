"""Worker of tests/test_gpu_two_ranks.py::test_peer_window_waits_are_bounded: two ranks on device 0 on the peer-window
transport; rank 1 never takes part in a reduction, so rank 0's all-reduce kernel must GIVE UP after its bounded wait
(5 s) and the library must report STORM_HIP_E_COMM -- not hang.  Rank 1 stays alive (its window mapped) until both are
done."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as td  # noqa: E402

from stormruler_amd import api, dist  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank = td.get_rank()
    ctx = api.Context(0)
    dist.connect_ipc(ctx)
    report = {"rank": rank}
    if rank == 0:
        a = api.DeviceVector(ctx, 1000)
        api.fill_with(a, 1.0)
        t = time.time()
        try:
            api.dot_product(a, a)  # a collective the other rank never joins
            ctx.sync()
            report["outcome"] = "returned"
        except api._lib.StormHipError as e:
            report["outcome"], report["status"], report["what"] = "error", e.status, str(e)
        report["seconds"] = time.time() - t
    else:
        time.sleep(9.0)
    td.barrier()
    with open(os.path.join(os.environ["STORM_REPORT_DIR"], f"rank{rank}.json"), "w") as f:
        json.dump(report, f)
    td.barrier()
    try:
        ctx.close()
    except Exception:
        pass
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
