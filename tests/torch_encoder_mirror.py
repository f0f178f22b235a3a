"""TEST INFRASTRUCTURE — plain-torch restatement of the encoder graphs (runs on CPU through ATen).

The product package holds the encoders as parameter containers (``oodgan.encoder``) with HIP forwards
(``oodgan.encoder_hip``) and no PyTorch compute path; these functions evaluate the same graphs with ``torch.nn.functional``
on a container's parameters, so that (a) the container's structure / key layout can be checked against the vectors the
reference produced without a GPU and (b) the HIP forwards have a second, independent reference beside the golden files.
Reference lines: psp_encoders.py:35-57,178-216; helpers.py:60-76,479-521; restyle_e4e_encoder.py:85-112;
feature_style_encoder.py:47-74; arcface/iresnet.py:46-61."""
import torch
import torch.nn.functional as F


def _bn(m, x):
    return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps)


def _conv(m, x):
    return F.conv2d(x, m.weight, m.bias, m.stride, m.padding)


def _se(m, x):
    g = x.mean(dim=(2, 3), keepdim=True)
    return x * torch.sigmoid(_conv(m.fc2, F.relu(_conv(m.fc1, g))))


def _bottleneck(m, x):
    """bottleneck_IR / bottleneck_IR_SE: BN -> conv3x3 -> PReLU -> conv3x3(stride) -> BN [-> SE] + shortcut."""
    r = m.res_layer
    y = _bn(r[4], _conv(r[3], F.prelu(_conv(r[1], _bn(r[0], x)), r[2].weight)))
    if len(r) > 5:
        y = _se(r[5], y)
    sc = m.shortcut_layer
    if isinstance(sc, torch.nn.MaxPool2d):
        st = sc.stride if isinstance(sc.stride, int) else sc.stride[0]
        short = x[:, :, ::st, ::st]
    else:
        short = _bn(sc[1], _conv(sc[0], x))
    return y + short


def _gradual_style(m, x):
    for layer in m.convs:
        x = _conv(layer, x) if isinstance(layer, torch.nn.Conv2d) else F.leaky_relu(x, 0.01)
    x = x.view(-1, m.out_c)
    return F.linear(x, m.linear.weight * m.linear.scale, bias=m.linear.bias * m.linear.lr_mul)


def _upsample_add(x, y):
    return F.interpolate(x, size=y.shape[-2:], mode='bicubic', align_corners=True) + y


def _input_layer(m, x):
    return F.prelu(_bn(m[1], _conv(m[0], x)), m[2].weight)


def encoder4editing_forward(enc, x, return_feats=False):
    x = _input_layer(enc.input_layer, x)
    feats = [x]
    c1 = c2 = c3 = None
    for i, layer in enumerate(enc.body):
        x = _bottleneck(layer, x)
        if i == 2:
            feats.append(x)
        if i == 6:
            c1 = x
            feats.append(x)
        elif i == 20:
            c2 = x
            feats.append(x)
        elif i == 23:
            c3 = x
            feats.append(x)
    w0 = _gradual_style(enc.styles[0], c3)
    w = w0.repeat(enc.style_count, 1, 1).permute(1, 0, 2)
    features, p2 = c3, None
    for i in range(1, min(enc.progressive_stage.value + 1, enc.style_count)):
        if i == enc.coarse_ind:
            p2 = _upsample_add(c3, _conv(enc.latlayer1, c2))
            features = p2
        elif i == enc.middle_ind:
            features = _upsample_add(p2, _conv(enc.latlayer2, c1))
        w[:, i] += _gradual_style(enc.styles[i], features)
    return (w, feats) if return_feats else w


def progressive_backbone_forward(enc, x, return_feats=False):
    x = _input_layer(enc.input_layer, x)
    feats = [x]
    for i, layer in enumerate(enc.body):
        x = _bottleneck(layer, x)
        if i in (2, 6, 20, 23):
            feats.append(x)
    w0 = _gradual_style(enc.styles[0], x)
    w = w0.repeat(enc.style_count, 1, 1).permute(1, 0, 2)
    for i in range(1, min(enc.progressive_stage.value + 1, enc.style_count)):
        w[:, i] += _gradual_style(enc.styles[i], x)
    return (w, feats) if return_feats else w


def _ibasic(m, x):
    out = _bn(m.bn3, _conv(m.conv2, F.prelu(_bn(m.bn2, _conv(m.conv1, _bn(m.bn1, x))), m.prelu.weight)))
    return out + (x if m.downsample is None else _bn(m.downsample[1], _conv(m.downsample[0], x)))


def fs_encoder_forward(enc, x, return_feats=False):
    def stage(blocks, t):
        for blk in blocks:
            t = _ibasic(blk, t)
        return t
    pool = lambda t: F.adaptive_avg_pool2d(t, (3, 3))
    x = _input_layer(enc.conv, x)
    taps, pooled = [x], []
    x = stage(enc.block_1, x)
    taps.append(x)
    pooled.append(pool(x))
    x = stage(enc.block_2, x)
    taps.append(x)
    pooled.append(pool(x))
    x = stage(enc.block_3, x)
    taps.append(x)
    c = enc.content_layer
    content = _bn(c[5], _conv(c[4], F.prelu(_bn(c[2], _conv(c[1], _bn(c[0], x))), c[3].weight)))
    pooled.append(pool(x))
    x = stage(enc.block_4, x)
    pooled.append(pool(x))
    d = torch.cat(pooled, dim=1).flatten(1)
    out = torch.stack([F.linear(d, s.weight, s.bias) for s in enc.styles], dim=1)
    return (out, content, taps) if return_feats else (out, content)
