"""The architecture-description language of the reference, as data.

The reference configures its networks with lists of ``(layer_name, config)``
tuples that ``build_sequential`` turns into ``torch.nn`` modules
(/root/reference/baryon_painter/models/utils.py:114-157) and offers four helper
functions that emit such lists (``conv_block``/``res_block``/``conv_down``/
``conv_up``, utils.py:40-112; used by scripts/CVAE_single_scale.py:104-133).
Here the same helpers (same names, same keyword arguments, same emitted tuples)
produce the descriptions; ``graph.py`` compiles them to HIP launch plans instead
of ``torch.nn`` modules.
"""

_ACTIVATIONS = {
    "relu": ("ReLU",),
    "prelu": ("prelu",),
    "tanh": ("tanh",),
    "sigmoid": ("sigmoid",),
    "softplus": ("softplus",),
}

# scale -> (kernel_size, padding, stride); utils.py:41-51
_SCALE_KPS = {2: (4, 1, 2), 4: (8, 2, 4)}


def conv_block(in_channel, out_channel, type="conv", scale=1, kernel=3, bias=False,
               batchnorm=True, activation="relu", relu_slope=0.2):
    """One (transposed) convolution [+ batchnorm] [+ activation].

    ``scale`` 1 keeps the resolution with an odd ``kernel`` and "same" padding;
    2 and 4 change it by that factor with (k,p,s) = (4,1,2) / (8,2,4)."""
    if scale == 1:
        if kernel % 2 != 1:
            raise ValueError("Kernel with scale=1 should be odd.")
        k, p, s = kernel, (kernel - 1) // 2, 1
    elif scale in _SCALE_KPS:
        k, p, s = _SCALE_KPS[scale]
    else:
        raise NotImplementedError("Scaling {} not supported yet!".format(scale))

    layers = [(type, {"in_channels": in_channel, "out_channels": out_channel,
                      "kernel_size": k, "padding": p, "stride": s, "bias": bias})]
    if batchnorm:
        layers.append(("batchnorm", {"num_features": out_channel}))

    act = None if activation is None else activation.lower()
    if act is None or act == "none":
        pass
    elif act == "leaky relu":
        layers.append(("Leaky ReLU", relu_slope))
    elif act in _ACTIVATIONS:
        layers.append(_ACTIVATIONS[act])
    else:
        raise NotImplementedError("Activation {} not supported yet!".format(activation))
    return layers


def res_block(n_channel):
    """Two 3x3 same-resolution convolutions with batchnorm, ReLU between them and
    after the skip addition (utils.py:79-98)."""
    def conv3():
        return ("conv", {"in_channels": n_channel, "out_channels": n_channel,
                         "kernel_size": 3, "padding": 1, "stride": 1, "bias": False})

    def bn():
        return ("batchnorm", {"num_features": n_channel})
    return ([conv3(), bn(), ("ReLU",), conv3(), bn()], ("ReLU",))


def _chain(in_channel, channels, scales, **kw):
    layers, c_in = [], in_channel
    for c_out, s in zip(channels, scales):
        layers += conv_block(in_channel=c_in, out_channel=c_out, scale=s, **kw)
        c_in = c_out
    return layers


def conv_down(in_channel, channels, scales, **kw_args):
    return _chain(in_channel, channels, scales, **kw_args)


def conv_up(in_channel, channels, scales, **kw_args):
    return _chain(in_channel, channels, scales, type="transp conv", **kw_args)


def fiducial_architecture(tile_size=512, predict_var=False, n_res=4):
    """The "CVAE fiducial" network: trained_models/CVAE/fiducial/architecture.txt
    (mean head only); ``predict_var=True`` gives the two-head superset that
    scripts/CVAE_single_scale.py:97-138 builds.  dim_z = tile_size/32."""
    n_scale, n_aux_label, n_x_feature = 1, 1, 1
    zs = tile_size // 32
    dim_z = (1, zs, zs)
    dim = (n_x_feature, tile_size, tile_size)

    def head(last_activation):
        return (conv_block(16, 8, kernel=7, bias=False, batchnorm=False, activation="PReLU")
                + conv_block(8, n_x_feature, kernel=5, bias=False, batchnorm=False, activation="PReLU")
                + conv_block(n_x_feature, n_x_feature, kernel=3, bias=False, batchnorm=False,
                             activation=last_activation))

    heads = (head("softplus"), head(None)) if predict_var else (head("softplus"),)
    arch = {
        "type": "Type-1",
        "dim_x": dim,
        "dim_y": (n_scale, tile_size, tile_size),
        "dim_z": dim_z,
        "n_x_features": n_x_feature,
        "aux_label": True,
        "prior_z_y": (conv_down(in_channel=1 + n_aux_label, channels=[8, 16, 32], scales=[2, 4, 4])
                      + conv_block(32, 2 * dim_z[0], kernel=5)
                      + [("unflatten", (2, *dim_z))]),
        "q_x_in": conv_down(in_channel=n_x_feature, channels=[8, 16, 32], scales=[2, 4, 4]),
        "q_y_in": conv_down(in_channel=1 + n_aux_label, channels=[8, 16, 32], scales=[2, 4, 4]),
        "q_x_y_out": conv_block(64, 2 * dim_z[0], kernel=5) + [("unflatten", (2, *dim_z))],
        "p_y_in": None,
        "p_z_in": conv_up(1, channels=[1, 1, 1], scales=[2, 4, 4], bias=False, batchnorm=True),
        "p_y_z_in": (conv_block(n_aux_label + n_scale + 1, 16, kernel=5)
                     + conv_down(in_channel=16, channels=[32, 64, 128], scales=[2, 2, 2])
                     + [("residual block", res_block(128)) for _ in range(n_res)]
                     + conv_up(128, channels=[64, 32, 16], scales=[2, 2, 2], bias=False,
                               batchnorm=True, activation="ReLU")),
        "p_y_z_out": heads,
        "min_x_var": 1e-7,
        "min_z_var": 1e-7,
        "L": 1,
    }
    return arch


# ----------------------------------------------------------------------------------- CGAN
# The reference repository documents its CGAN but holds no code for it (SURVEY.md key fact 2):
# the layer tables below restate trained_models/README.md:106-128 and the pickled generator
# structure (rows g1, g2 of SURVEY.md 8a) in the same architecture language.
def leaky_res_block(n_channel, slope=0.2):
    """``res_block`` with LeakyReLU instead of ReLU (generator's residual blocks, README.md:121)."""
    body, _ = res_block(n_channel)
    body[2] = ("Leaky ReLU", slope)
    return (body, ("Leaky ReLU", slope))


def cgan_generator_architecture(n_res=9, slope=0.2):
    """``resnet_translator``: k9 stem, two stride-2 encoders, n_res residual blocks, two
    transposed-conv decoders (output_padding 1), k9 head + Tanh."""
    def cbl(kind, cin, cout, k, s, p, bias, **extra):
        cfg = {"in_channels": cin, "out_channels": cout, "kernel_size": k, "padding": p, "stride": s, "bias": bias}
        cfg.update(extra)
        return [(kind, cfg), ("batchnorm", {"num_features": cout}), ("Leaky ReLU", slope)]
    layers = cbl("conv", 2, 32, 9, 1, 4, False) + cbl("conv", 32, 64, 3, 2, 1, True) + cbl("conv", 64, 128, 3, 2, 1, True)
    layers += [("residual block", leaky_res_block(128, slope)) for _ in range(n_res)]
    layers += cbl("transp conv", 128, 64, 3, 2, 1, True, output_padding=1)
    layers += cbl("transp conv", 64, 32, 3, 2, 1, True, output_padding=1)
    layers += [("conv", {"in_channels": 32, "out_channels": 1, "kernel_size": 9, "padding": 4, "stride": 1,
                         "bias": True}), ("tanh",)]
    return layers


def cgan_discriminator_architecture(slope=0.2):
    """pix2pix-style PatchGAN with spectral normalisation on every layer, no batch-norm
    (README.md:101,106-114); padding 1 assumed (unspecified in the reference)."""
    def sn(cin, cout, s, bias):
        return ("sn conv", {"in_channels": cin, "out_channels": cout, "kernel_size": 4, "padding": 1, "stride": s,
                            "bias": bias})
    return [sn(3, 64, 2, True), ("Leaky ReLU", slope), sn(64, 128, 2, True), ("Leaky ReLU", slope),
            sn(128, 256, 2, False), ("Leaky ReLU", slope), sn(256, 512, 1, True), ("Leaky ReLU", slope),
            sn(512, 1, 1, True), ("sigmoid",)]
