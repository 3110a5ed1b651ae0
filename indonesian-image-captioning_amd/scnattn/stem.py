"""The stem of the ResNet-152 trunk (children 0..3: conv1 7x7/2, bn1, relu, maxpool 3x3/2) on the hand-written
kernels of csrc/stem.hip.  Reference: `nn.Sequential(*list(resnet152.children())[:-2])`, models/encoders/caption.py:17-22
(and tagger.py:18-24); `fine_tune` keeps these four children frozen in every configuration (caption.py:46-57), so the
fused path is forward-only -- a stem that must produce gradients takes the plain module path instead.

    z  = conv7(x) (+ per-workgroup sums for bn1's batch statistics)       scnattn_stem_conv7
    bn1 statistics / running-stat update / folded {scale, shift}           scnattn_bn_finalize
    out = maxpool(relu(z * scale + shift))                                 scnattn_stem_bn_relu_maxpool
The image batch is read in whatever memory format it arrives in (strides are passed), the output is channels-last."""
import torch

from . import _lib
from . import conv as _conv


def usable(trunk, x):
    if not (_conv.ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") != torch.bfloat16:
        return False
    if len(trunk) < 4:
        return False
    c1, bn, pool = trunk[0], trunk[1], trunk[3]
    if not (isinstance(c1, torch.nn.Conv2d) and isinstance(bn, torch.nn.BatchNorm2d) and isinstance(pool, torch.nn.MaxPool2d)):
        return False
    if c1.weight.shape != (64, 3, 7, 7) or c1.stride != (2, 2) or c1.padding != (3, 3) or c1.dilation != (1, 1) \
            or c1.bias is not None or c1.groups != 1 or c1.weight.dtype != torch.float32 or x.shape[1] != 3:
        return False
    if bn.weight is None or bn.running_mean is None or (bn.training and bn.momentum is None):
        return False
    ks = pool.kernel_size if isinstance(pool.kernel_size, tuple) else (pool.kernel_size,) * 2
    sd = pool.stride if isinstance(pool.stride, tuple) else (pool.stride,) * 2
    pd = pool.padding if isinstance(pool.padding, tuple) else (pool.padding,) * 2
    if ks != (3, 3) or sd != (2, 2) or pd != (1, 1) or pool.dilation not in (1, (1, 1)) or pool.ceil_mode:
        return False
    # forward-only kernels: nothing here may need a gradient
    if torch.is_grad_enabled() and (x.requires_grad or c1.weight.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad):
        return False
    return True


def stem(trunk, x, bf16=False):
    """(N,3,H,W) -> (N,64,Hp,Wp) channels-last, Hp = ((H-1)//2) // 2 + 1 (= H/4 for the 256 x 256 inputs).  bf16: the
    pooled map is written as bf16 (the mixed-precision trunk); the convolution itself stays fp32 (K = 147: 1 % of the
    trunk's work)."""
    c1, bn = trunk[0], trunk[1]
    h = _lib.lib()
    dev = x.device
    st = torch._C._cuda_getCurrentRawStream(dev.index)
    N, _, H, W = x.shape
    Hz, Wz = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hz - 1) // 2 + 1, (Wz - 1) // 2 + 1
    w = c1.weight
    z = torch.empty((N * Hz * Wz, 64), device=dev, dtype=torch.float32)
    ss = torch.empty((64, 2), device=dev, dtype=torch.float32)
    if bn.training:
        if not getattr(bn, "counter_managed", False) and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        nt = h.scnattn_stem_tiles(N, H, W)
        ldp = (nt + 3) & ~3
        part = torch.empty((2, 64, ldp), device=dev, dtype=torch.float32)     # channel-major partials, one entry per workgroup
        stats = torch.empty((2, 64), device=dev, dtype=torch.float32)
        shift = _conv._shift(bn)
        _conv._chk(h.scnattn_stem_conv7(st, N, H, W, x.data_ptr(), *x.stride(), w.data_ptr(), *w.stride(), z.data_ptr(),
                                        part.data_ptr(), shift.data_ptr()), "scnattn_stem_conv7")
        _conv._chk(h.scnattn_bn_finalize(st, N * Hz * Wz, 64, part.data_ptr(), ldp, nt, shift.data_ptr(), bn.eps,
                                         bn.momentum, stats[0].data_ptr(), stats[1].data_ptr(),
                                         bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.weight.data_ptr(),
                                         bn.bias.data_ptr(), ss.data_ptr()), "scnattn_bn_finalize")
        bn._scn_shift = stats[0]
    else:
        _conv._chk(h.scnattn_stem_conv7(st, N, H, W, x.data_ptr(), *x.stride(), w.data_ptr(), *w.stride(), z.data_ptr(),
                                        None, None), "scnattn_stem_conv7")
        with torch.no_grad():       # 64 channels: the folded {scale, shift} of the running statistics
            sc = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            ss.copy_(torch.stack([sc, bn.bias - bn.running_mean * sc], dim=1))
    out = torch.empty((N * Hp * Wp, 64), device=dev, dtype=torch.bfloat16 if bf16 else torch.float32)
    _conv._chk(h.scnattn_stem_bn_relu_maxpool(st, N, Hz, Wz, 64, z.data_ptr(), ss.data_ptr(), out.data_ptr(), 1 if bf16 else 0),
               "scnattn_stem_bn_relu_maxpool")
    return out.view(N, Hp, Wp, 64).permute(0, 3, 1, 2)


def run_trunk(trunk, x):
    """`trunk(x)` with the four stem children on the fused kernels when they qualify."""
    if usable(trunk, x):
        bf16 = torch.is_autocast_enabled()          # bf16 autocast = the mixed-precision trunk (scnattn/conv16.py)
        if bf16:
            from . import conv16 as _c16
            _c16.refresh_weights(trunk)              # fp32 master weights -> bf16 operand copies, one launch
        y = stem(trunk, x, bf16)
        for child in list(trunk.children())[4:]:
            y = child(y)
        return y
    return trunk(x)
