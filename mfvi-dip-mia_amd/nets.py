"""The skip() hour-glass family of the reference as plain torch modules (structure + parameter holders).

Same constructor surface and — because checkpoints and the notebooks address layers by name — the same module
names / state_dict keys as models/skip.py:5-134, models/common.py:15-135 and models/__init__.py:4-27 of the
reference.  The modules are executable on their own (plain torch), but the product path never runs them: MeanFieldVI
(bayes.py) compiles the tree into a layer program for libmfvi_hip.
"""
import torch
import torch.nn as nn


class Concat(nn.Module):
    """Runs every child on the same input and concatenates along `dim`, centre-cropping to the smallest spatial size
    (models/common.py:15-46)."""

    def __init__(self, dim, *children_):
        super().__init__()
        self.dim = dim
        for i, m in enumerate(children_):
            self.add_module(str(i), m)

    def forward(self, x):
        outs = [m(x) for m in self._modules.values()]
        h = min(o.shape[2] for o in outs); w = min(o.shape[3] for o in outs)
        cropped = []
        for o in outs:
            dh, dw = (o.shape[2] - h) // 2, (o.shape[3] - w) // 2
            cropped.append(o[:, :, dh:dh + h, dw:dw + w])
        return torch.cat(cropped, dim=self.dim)

    def __len__(self):
        return len(self._modules)


def _act(act_fun):
    if act_fun == 'LeakyReLU':
        return nn.LeakyReLU(0.2, inplace=True)          # models/common.py:83
    raise NotImplementedError("act_fun=%r: only 'LeakyReLU' (the MFVI runners' choice) is built" % (act_fun,))


def _conv_block(cin, cout, k, stride, bias, pad, downsample_mode, tag, number, dropout_mode='None', dropout_p=0.5):
    """[ReflectionPad2d(k//2)] + Conv2d(padding=0) [+ Dropout2d(p)], children named '<Class>_<tag>_<number>'
    (models/common.py:100-135; the dropout layers belong to the MC-dropout runners' nets)."""
    if stride != 1 and downsample_mode != 'stride':
        raise NotImplementedError("downsample_mode=%r: only 'stride' is built" % (downsample_mode,))
    if pad != 'reflection':
        raise NotImplementedError("pad=%r: only 'reflection' is built" % (pad,))
    seq = nn.Sequential()
    seq.add_module('ReflectionPad2d_%s_%d' % (tag, number), nn.ReflectionPad2d((k - 1) // 2))
    seq.add_module('Conv2d_%s_%d' % (tag, number), nn.Conv2d(cin, cout, k, stride, padding=0, bias=bias))
    if dropout_mode == '2d':
        seq.add_module('Dropout2d_%s_%d' % (tag, number), nn.Dropout2d(p=dropout_p))
    elif dropout_mode not in ('None', None):
        raise NotImplementedError("dropout_mode=%r: only '2d' (the MC-dropout runners' setting) is built" % (dropout_mode,))
    return seq


def skip(num_input_channels=2, num_output_channels=3,
         num_channels_down=(16, 32, 64, 128, 128), num_channels_up=(16, 32, 64, 128, 128), num_channels_skip=(4, 4, 4, 4, 4),
         filter_size_down=3, filter_size_up=3, filter_skip_size=1, need_sigmoid=True, need_bias=True,
         pad='zero', upsample_mode='nearest', downsample_mode='stride', act_fun='LeakyReLU', need1x1_up=True,
         dropout_mode_down='2d', dropout_p_down=0.5, dropout_mode_up='2d', dropout_p_up=0.5,
         dropout_mode_skip='None', dropout_p_skip=0.5, dropout_mode_output='None', dropout_p_output=0.5):
    """Encoder-decoder with skip connections (defaults as models/skip.py:5-14, including its '2d' dropout after the deeper / up
    convolutions: get_net() and the MFVI / DIP / SGLD runners pass 'None', the MC-dropout runners '2d').  Names are assigned directly (the reference renames after the fact):
    scale i holds  Concat_up_n{0: skip branch, 1: deeper branch}, BatchNorm2d_up_n, Sequential_up_n, BatchNorm2d_up_n_1,
    LeakyReLU_up_n, Sequential_up_{n+1}, BatchNorm2d_up_{n+1}, LeakyReLU_up_{n+1}  with n = 2*(n_scales-i)-1."""
    n_scales = len(num_channels_down)
    assert len(num_channels_up) == n_scales and len(num_channels_skip) == n_scales
    if dropout_mode_output not in ('None', None):
        raise NotImplementedError("dropout on the output convolution (no BatchNorm behind it) is not built; no runner of the reference uses it")
    drop_down, drop_up, drop_skip = (dropout_mode_down, dropout_p_down), (dropout_mode_up, dropout_p_up), (dropout_mode_skip, dropout_p_skip)
    if need_sigmoid:
        raise NotImplementedError("need_sigmoid=True is not used by the MFVI runners (models/__init__.py:4)")
    step = 2 if need1x1_up else 1          # 'up' convolutions per scale: the reference's running counter for the names

    def per_scale(v):
        return list(v) if isinstance(v, (list, tuple)) else [v] * n_scales
    up_modes, down_modes = per_scale(upsample_mode), per_scale(downsample_mode)
    fdown, fup = per_scale(filter_size_down), per_scale(filter_size_up)

    def add(seq, name, module):
        """add_module with the reference's collision rule: a second 'X_up_n' in the same container becomes 'X_up_n_1'."""
        if name in seq._modules:
            name += '_1'
        seq.add_module(name, module)

    def build_scale(i, cin, seq):
        """Fill `seq` (the reference's model_tmp) with scale i; returns nothing."""
        n_up = step * (n_scales - i) - 1                   # 9, 7, 5, 3, 1  (need1x1_up=False: 4, 3, 2, 1, 0)
        n_deep = 2 * i + 1                                 # 1, 3, 5, 7, 9
        ns, nd, nu = num_channels_skip[i], num_channels_down[i], num_channels_up[i]
        dp = nn.Sequential()
        dp.add_module('Sequential_deeper_%d' % n_deep, _conv_block(cin, nd, fdown[i], 2, need_bias, pad, down_modes[i], 'deeper', n_deep, *drop_down))
        dp.add_module('BatchNorm2d_deeper_%d' % n_deep, nn.BatchNorm2d(nd))
        dp.add_module('LeakyReLU_deeper_%d' % n_deep, _act(act_fun))
        dp.add_module('Sequential_deeper_%d' % (n_deep + 1), _conv_block(nd, nd, fdown[i], 1, need_bias, pad, 'stride', 'deeper', n_deep + 1, *drop_down))
        dp.add_module('BatchNorm2d_deeper_%d' % (n_deep + 1), nn.BatchNorm2d(nd))
        dp.add_module('LeakyReLU_deeper_%d' % (n_deep + 1), _act(act_fun))
        if i < n_scales - 1:
            inner = nn.Sequential()
            build_scale(i + 1, nd, inner)
            dp.add_module('7', inner)                     # the reference's un-renamed positional keys
            dp.add_module('8', nn.Upsample(scale_factor=2, mode=up_modes[i]))
            k_in = num_channels_up[i + 1]
        else:
            dp.add_module('7', nn.Upsample(scale_factor=2, mode=up_modes[i]))
            k_in = nd
        if ns != 0:
            sk = nn.Sequential()
            sk.add_module('Sequential_skip_%d' % (i + 1), _conv_block(cin, ns, filter_skip_size, 1, need_bias, pad, 'stride', 'skip', i + 1, *drop_skip))
            sk.add_module('BatchNorm2d_skip_%d' % (i + 1), nn.BatchNorm2d(ns))
            sk.add_module('LeakyReLU_skip_%d' % (i + 1), _act(act_fun))
            add(seq, 'Concat_up_%d' % n_up, Concat(1, sk, dp))
        else:                                              # models/skip.py:62-66: no skip branch, the deeper path is added as is
            add(seq, 'Sequential_up_%d' % n_up, dp)
        add(seq, 'BatchNorm2d_up_%d' % n_up, nn.BatchNorm2d(ns + k_in))
        add(seq, 'Sequential_up_%d' % n_up, _conv_block(ns + k_in, nu, fup[i], 1, need_bias, pad, 'stride', 'up', n_up, *drop_up))
        add(seq, 'BatchNorm2d_up_%d' % n_up, nn.BatchNorm2d(nu))
        add(seq, 'LeakyReLU_up_%d' % n_up, _act(act_fun))
        if need1x1_up:
            add(seq, 'Sequential_up_%d' % (n_up + 1), _conv_block(nu, nu, 1, 1, need_bias, pad, 'stride', 'up', n_up + 1, *drop_up))
            add(seq, 'BatchNorm2d_up_%d' % (n_up + 1), nn.BatchNorm2d(nu))
            add(seq, 'LeakyReLU_up_%d' % (n_up + 1), _act(act_fun))

    model = nn.Sequential()
    build_scale(0, num_input_channels, model)
    model.add_module(str(len(model) + 1), _conv_block(num_channels_up[0], num_output_channels, 1, 1, need_bias, pad, 'stride', 'up', step * n_scales + 1))
    return model


def get_net(input_depth, NET_TYPE, pad, upsample_mode, n_channels=3, act_fun='LeakyReLU', need_sigmoid=False,
            skip_n33d=128, skip_n33u=128, skip_n11=4, num_scales=5, downsample_mode='stride',
            dropout_mode_down='None', dropout_p_down=0.5, dropout_mode_up='None', dropout_p_up=0.5,
            dropout_mode_skip='None', dropout_p_skip=0.5, dropout_mode_output='None', dropout_p_output=0.5):
    """models/__init__.py:4-27 (its dropout defaults are 'None', unlike skip()'s)."""
    if NET_TYPE != 'skip':
        raise NotImplementedError("NET_TYPE=%r: only 'skip' exists in the reference" % (NET_TYPE,))

    def lst(v):
        return [v] * num_scales if isinstance(v, int) else list(v)
    return skip(input_depth, n_channels, num_channels_down=lst(skip_n33d), num_channels_up=lst(skip_n33u),
                num_channels_skip=lst(skip_n11), upsample_mode=upsample_mode, downsample_mode=downsample_mode,
                need_sigmoid=need_sigmoid, need_bias=True, pad=pad, act_fun=act_fun,
                dropout_mode_down=dropout_mode_down, dropout_p_down=dropout_p_down, dropout_mode_up=dropout_mode_up, dropout_p_up=dropout_p_up,
                dropout_mode_skip=dropout_mode_skip, dropout_p_skip=dropout_p_skip, dropout_mode_output=dropout_mode_output,
                dropout_p_output=dropout_p_output)
