"""Streaming-read reference point: gmpi_rgba_range_check_launch over a 3.2 GB bf16 volume (run under rocprofv3 for TCP counters)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ml_gmpi_amd import _lib
lib = _lib.load_library()
dev = torch.device("cuda")
x = torch.rand((4, 96, 4, 1024, 1024), device=dev).to(torch.bfloat16)
status = torch.zeros(4, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def f(): _lib.check(lib.gmpi_rgba_range_check_launch(x.data_ptr(), 1, x.numel(), status.data_ptr(), st), "rc")
for _ in range(2): f()
torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(8): f()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 8
print(f"stream read {x.numel()*2/1e9:.2f} GB in {ms:.3f} ms = {x.numel()*2/ms/1e9:.2f} TB/s")
