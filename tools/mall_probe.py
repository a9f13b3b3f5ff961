"""Read-latency calibration for the HBM / Infinity-Cache split of FETCH_SIZE: the product's streaming-read probe kernel
(gmpi_stream_probe_launch) over (a) a buffer that fits the 256 MB Infinity Cache, read again and again (every L2 miss after the first pass is a
MALL hit), (b) a buffer far larger than it (every L2 miss goes to HBM).  Run under
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum
and compare LEVEL / RDREQ (average latency of an L2 -> fabric read, in L2 clocks) with the same ratio of the render kernel.
usage: python tools/mall_probe.py <MiB> <launches>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ml_gmpi_amd import _lib  # noqa: E402

mib, n = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.load_library()
dev = torch.device("cuda:0")
buf = torch.empty(mib << 20, dtype=torch.uint8, device=dev)
buf.zero_()
st = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
cs = torch.cuda.current_stream(dev).cuda_stream
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    _lib.check(lib.gmpi_stream_probe_launch(buf.data_ptr(), buf.numel(), st.data_ptr(), cs), "probe")
ev0.record()
for _ in range(n):
    _lib.check(lib.gmpi_stream_probe_launch(buf.data_ptr(), buf.numel(), st.data_ptr(), cs), "probe")
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / n
print(f"stream probe over {mib} MiB: {ms:.4f} ms per pass = {buf.numel() / ms / 1e6:.0f} GB/s")
