import sys, torch
sys.path.insert(0, '.')
from txt2vid_amd import functional as TF
orig = TF.ConvG.backward
def bw(ctx, *gys):
    print('ConvG.backward w', tuple(ctx.saved_tensors[0].shape), 'gys:', [g is not None for g in gys], 'needs', ctx.needs_input_grad[3:])
    return orig(ctx, *gys)
TF.ConvG.backward = staticmethod(bw)
w1 = torch.nn.Parameter(torch.randn(8, 4, 3, 3, 3, device='cuda') * 0.1)
w2 = torch.nn.Parameter(torch.randn(6, 8, 3, 3, 3, device='cuda') * 0.1)
a = torch.randn(2, 4, 2, 4, 4, device='cuda')
b = torch.randn(2, 4, 2, 4, 4, device='cuda', requires_grad=True)
h = TF.conv_group([a, b], w1, None)
h = [TF.add(t, t) for t in h]
y = TF.conv_group(h, w2, None, relu_in=True)
y = [TF.avg_pool3d(t, (1, 2, 2), (2, 2, 2)) for t in y]
print('--- grad wrt b of yb only')
g, = torch.autograd.grad(y[1].sum(), b, create_graph=True)
