"""ATen ops (count, GPU time, issuing frame) of one Resnet3D forward + backward pass as `d_fwdbwd_roofline` runs it (developer tool)."""
import sys, os, collections, torch
sys.path.insert(0, '/root/repo')
from txt2vid_amd import functional as TF
from txt2vid_amd.models.resnet3d import Resnet3D
from txt2vid_amd.util.torch.init import init
from txt2vid_amd.dist import model_arena
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
torch.manual_seed(100)
D = Resnet3D(num_channels=1, with_attn=True); init(D, 'xavier'); D.to(dev)
x = (torch.rand(32, 1, 16, 64, 64) * 2 - 1).to(dev)
sink = TF.GradSink([model_arena(D, TF.copy_into)]); TF.set_grad_sink(sink)
def fb():
    for p in D.parameters(): p.grad = None
    TF.grad_sink_reset(); u, _, _ = D(x=x); TF.vec_sum(u.reshape(-1)).backward(); TF.grad_sink_flush()
for _ in range(2): fb()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fb(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=15):
    if e.key.startswith('aten::') and getattr(e, 'device_time_total', 0) > 0:
        fr = [f for f in (e.stack or []) if 'txt2vid_amd' in f]
        rows.append((e.count, e.key, e.device_time_total, (fr[0] if fr else '-').split('/root/repo/')[-1][:100]))
for r in sorted(rows, key=lambda r: -r[2]): print('%3d %-22s %8.1f us  %s' % r)
