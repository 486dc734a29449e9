"""Is a large->scalar torch reduction (multi-block, semaphore scratch zeroed by hipMemsetAsync) replay-safe in a hipGraph?"""
import torch
x = torch.randn(6, 64, 200, 200, device='cuda')
junk = None
def body(two_stage):
    y = (x * 2.0)
    big = torch.empty(40 << 20, device='cuda').fill_(3.0)     # churn the pool so scratch lands on dirty memory
    del big
    if two_stage:
        return y.view(-1, 4096).sum(1).sum() / y.numel()
    return y.mean()
for two_stage in (False, True):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): ref = body(two_stage)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body(two_stage)
    vals = []
    for i in range(4):
        g.replay(); torch.cuda.synchronize(); vals.append(float(out))
    print(f'two_stage={two_stage} eager={float(ref):.6f} replays={vals}', flush=True)
