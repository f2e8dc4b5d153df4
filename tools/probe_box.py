"""One-off GPU-box probe: device, host cores, MIOpen GRU fwd/bwd with dropout."""
import os, time, torch
print("nproc", os.cpu_count(), "torch", torch.__version__, "hip", torch.version.hip)
print("dev", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0))
g = torch.nn.GRU(200, 100, num_layers=2, bidirectional=True, dropout=0.5).cuda()
x = torch.randn(110, 96, 200, device="cuda", requires_grad=True)
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    y, _ = g(x); y.sum().backward(); torch.cuda.synchronize()
    print("gru fwd+bwd ms", (time.time() - t) * 1e3)
gc = torch.nn.GRU(200, 100, num_layers=2, bidirectional=True).cuda().eval()
import copy
gcpu = copy.deepcopy(gc).cpu()
xe = torch.randn(110, 8, 200)
print("gru gpu-vs-cpu maxdiff", (gc(xe.cuda())[0].cpu() - gcpu(xe)[0]).abs().max().item())
