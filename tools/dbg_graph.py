"""Development: the fused causal step (persistent launch, dynamic hand-out, memset node) under HIP graph capture, workspace owned by the script."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ab import load, Variant
B, H, S, D = (int(x) for x in os.environ.get("SHAPE", "4,8,4096,128").split(","))
L = load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quantumattention_amd", "libqattn_hip.so"))
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    x = Variant("g", L, q, k, v, True, 0)
    x.fused(0); x.fused(0)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
eager = x.out.clone()
qws = L.qattn_quant_qkv_workspace_bytes(B, H, H)
off = (qws + 15) // 16 * 16
print("workspace bytes", x.ws_f.numel(), "sched words after eager:", x.ws_f[off:off + 32].view(torch.int32).tolist(), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    x.st = torch.cuda.current_stream().cuda_stream
    x.fused(0)
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(3):
    x.out.zero_()
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, "equal to eager:", torch.equal(x.out, eager), "max diff", (x.out.float() - eager.float()).abs().max().item(),
          "sched words:", x.ws_f[off:off + 32].view(torch.int32).tolist(), flush=True)
