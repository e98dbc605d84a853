"""TEST INFRASTRUCTURE ONLY — functional fp32 restatement of the KL-VAE decode paths.

Follows lvdm/modules/networks/ae_modules.py (Decoder.forward :661-702, ResnetBlock.forward :228-248,
AttnBlock.forward :53-78, Upsample.forward :123-127, Normalize :15-16 = GroupNorm(32, eps 1e-6), swish :10-12),
lvdm/models/autoencoder_adaptor.py (VAEDecoderadaptor.forward :277-317) and lvdm/models/autoencoder.py
(decode :136-139, decode_with_conf_adaptor :120-127). Operates on a reference-format AutoencoderKL state_dict.
Encode side: Encoder.forward ae_modules.py:537-580, Downsample.forward :102-109,
VAEEncoderadaptor.forward autoencoder_adaptor.py:166-199, AutoencoderKL.encode / encode_with_adaptor autoencoder.py:104-134;
returns the posterior's `parameters` (mean | logvar), i.e. quant_conv(encoder(x)).
"""
import torch
import torch.nn.functional as F


def _gn(sd, x, p):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def _conv(sd, x, p, pad):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=pad)


def _swish(x):
    return x * torch.sigmoid(x)


def _resnet(sd, x, p):
    h = _conv(sd, _swish(_gn(sd, x, p + ".norm1")), p + ".conv1", 1)
    h = _conv(sd, _swish(_gn(sd, h, p + ".norm2")), p + ".conv2", 1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, x, p + ".nin_shortcut", 0)
    return x + h


def _attn(sd, x, p):
    b, c, hh, ww = x.shape
    h = _gn(sd, x, p + ".norm")
    q, k, v = (_conv(sd, h, f"{p}.{n}", 0).reshape(b, c, hh * ww) for n in ("q", "k", "v"))
    w = torch.softmax(torch.bmm(q.permute(0, 2, 1), k) * (int(c) ** -0.5), dim=2)   # [b, i, j]
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)                       # sum_j v[c, j] w[i, j]
    return x + _conv(sd, h, p + ".proj_out", 0)


def decoder_features(sd, ddconfig, z, prefix="decoder"):
    """Decoder up to (not including) norm_out: the 'give_pre_end' tensor."""
    nres, nlev = ddconfig["num_res_blocks"], len(ddconfig["ch_mult"])
    h = _conv(sd, z, prefix + ".conv_in", 1)
    h = _resnet(sd, h, prefix + ".mid.block_1")
    h = _attn(sd, h, prefix + ".mid.attn_1")
    h = _resnet(sd, h, prefix + ".mid.block_2")
    for lvl in reversed(range(nlev)):
        for blk in range(nres + 1):
            h = _resnet(sd, h, f"{prefix}.up.{lvl}.block.{blk}")
        if lvl != 0:
            h = _conv(sd, F.interpolate(h, scale_factor=2.0, mode="nearest"), f"{prefix}.up.{lvl}.upsample.conv", 1)
    return h


def _head(sd, h, prefix):
    return _conv(sd, _swish(_gn(sd, h, prefix + ".norm_out")), prefix + ".conv_out", 1)


@torch.no_grad()
def decode(sd, ddconfig, z):
    """AutoencoderKL.decode: z [n,4,h,w] (already divided by scale_factor) -> [n,3,8h,8w]."""
    z = _conv(sd, z, "post_quant_conv", 0)
    return _head(sd, decoder_features(sd, ddconfig, z), "decoder")


@torch.no_grad()
def decode_with_conf_adaptor(sd, ddconfig, adaptorconfig, z):
    """AutoencoderKL.decode_with_conf_adaptor: -> [n, 3 + 1, 8h, 8w] (xyz + confidence logit)."""
    z = _conv(sd, z, "post_quant_conv", 0)
    feat = decoder_features(sd, ddconfig, z)
    rgb = _head(sd, feat, "decoder")
    h = feat
    for blk in range(adaptorconfig["num_res_blocks"] + 1):
        h = _resnet(sd, h, f"decoder_adaptor.up.0.block.{blk}")
    return torch.cat([rgb, _head(sd, h, "decoder_adaptor")], dim=1)


def encoder_forward(sd, ddconfig, x, prefix="encoder"):
    nres, nlev = ddconfig["num_res_blocks"], len(ddconfig["ch_mult"])
    h = _conv(sd, x, prefix + ".conv_in", 1)
    for lvl in range(nlev):
        for blk in range(nres):
            h = _resnet(sd, h, f"{prefix}.down.{lvl}.block.{blk}")
        if lvl != nlev - 1:
            p = f"{prefix}.down.{lvl}.downsample.conv"
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[p + ".weight"], sd[p + ".bias"], stride=2)
    h = _resnet(sd, h, prefix + ".mid.block_1")
    h = _attn(sd, h, prefix + ".mid.attn_1")
    h = _resnet(sd, h, prefix + ".mid.block_2")
    return _head(sd, h, prefix)


@torch.no_grad()
def encode(sd, ddconfig, x):
    """AutoencoderKL.encode(x).parameters: x [n,3,H,W] -> moments [n, 2*embed_dim, H/8, W/8]."""
    return _conv(sd, encoder_forward(sd, ddconfig, x), "quant_conv", 0)


@torch.no_grad()
def encode_with_adaptor(sd, ddconfig, adaptorconfig, x):
    h = _conv(sd, x, "encoder_adaptor.conv_in", 1)
    for blk in range(adaptorconfig["num_res_blocks"]):
        h = _resnet(sd, h, f"encoder_adaptor.down.0.block.{blk}")
    x = x + _head(sd, h, "encoder_adaptor")
    return _conv(sd, encoder_forward(sd, ddconfig, x), "quant_conv", 0)


def first_stage_encoding(moments, scale_factor, noise):
    """DiagonalGaussianDistribution.sample + get_first_stage_encoding (distributions.py:24-40, ddpm3d.py:674-681)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return scale_factor * (mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise)
