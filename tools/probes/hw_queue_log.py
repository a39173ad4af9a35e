"""Which hardware queue does each stream of a process get?  Run under AMD_LOG_LEVEL=4 and grep the runtime's queue messages (ROCclr prints the HSA queue
it creates or re-uses for every stream).  Marks in the log (`### ...` on stderr) say which Python statement the following runtime lines belong to."""
import sys
import torch

def mark(s):
    sys.stderr.write(f"### {s}\n"); sys.stderr.flush()

dev = torch.device("cuda:0")
mark("first kernel on the null stream")
x = torch.zeros(1, device=dev); torch.cuda.synchronize()
mark("high-priority stream created")
hi = torch.cuda.Stream(device=dev, priority=-1)
mark("high-priority stream first use")
with torch.cuda.stream(hi):
    torch.zeros(1, device=dev)
torch.cuda.synchronize()
ss = []
for i in range(6):
    mark(f"normal stream {i} created")
    s = torch.cuda.Stream(device=dev)
    ss.append(s)
    mark(f"normal stream {i} first use")
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
    torch.cuda.synchronize()
mark("graph capture on normal stream 0")
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(ss[0]):
    y = torch.zeros(8, device=dev)
    with torch.cuda.graph(g, stream=ss[0]):
        y.add_(1)
mark("graph replay on the null stream")
g.replay(); torch.cuda.synchronize()
mark("done")
