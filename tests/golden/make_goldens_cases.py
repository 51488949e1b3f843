"""Operator-level golden cases (data only; shared by make_goldens.py and the tests)."""
# ------------------------------------------------------------------ op-level cases
OP_CASES = [
    # name, layer type, cfg, (N,H,W)
    ("conv_k4s2_1_8",   "conv", dict(in_channels=1, out_channels=8, kernel_size=4, padding=1, stride=2), (2, 32, 32)),
    ("conv_k4s2_2_8",   "conv", dict(in_channels=2, out_channels=8, kernel_size=4, padding=1, stride=2), (2, 32, 32)),
    ("conv_k8s4_8_16",  "conv", dict(in_channels=8, out_channels=16, kernel_size=8, padding=2, stride=4), (2, 32, 32)),
    ("conv_k8s4_16_32", "conv", dict(in_channels=16, out_channels=32, kernel_size=8, padding=2, stride=4), (2, 16, 16)),
    ("conv_k5s1_64_2",  "conv", dict(in_channels=64, out_channels=2, kernel_size=5, padding=2, stride=1), (2, 8, 8)),
    ("conv_k5s1_32_2",  "conv", dict(in_channels=32, out_channels=2, kernel_size=5, padding=2, stride=1), (3, 4, 4)),
    ("conv_k5s1_3_16",  "conv", dict(in_channels=3, out_channels=16, kernel_size=5, padding=2, stride=1), (2, 24, 40)),
    ("conv_k4s2_16_32", "conv", dict(in_channels=16, out_channels=32, kernel_size=4, padding=1, stride=2), (2, 32, 32)),
    ("conv_k4s2_32_64", "conv", dict(in_channels=32, out_channels=64, kernel_size=4, padding=1, stride=2), (2, 16, 32)),
    ("conv_k4s2_64_128", "conv", dict(in_channels=64, out_channels=128, kernel_size=4, padding=1, stride=2), (2, 16, 16)),
    ("conv_k3s1_128_128", "conv", dict(in_channels=128, out_channels=128, kernel_size=3, padding=1, stride=1), (2, 8, 16)),
    ("conv_k3s1_1_1",   "conv", dict(in_channels=1, out_channels=1, kernel_size=3, padding=1, stride=1), (2, 32, 32)),
    ("conv_k7s1_16_8",  "conv", dict(in_channels=16, out_channels=8, kernel_size=7, padding=3, stride=1), (2, 24, 32)),
    ("conv_k5s1_8_1",   "conv", dict(in_channels=8, out_channels=1, kernel_size=5, padding=2, stride=1), (2, 32, 32)),
    ("convT_k4s2_1_1",  "transp conv", dict(in_channels=1, out_channels=1, kernel_size=4, padding=1, stride=2), (2, 8, 8)),
    ("convT_k8s4_1_1",  "transp conv", dict(in_channels=1, out_channels=1, kernel_size=8, padding=2, stride=4), (2, 8, 8)),
    ("convT_k4s2_128_64", "transp conv", dict(in_channels=128, out_channels=64, kernel_size=4, padding=1, stride=2), (2, 8, 8)),
    ("convT_k4s2_64_32", "transp conv", dict(in_channels=64, out_channels=32, kernel_size=4, padding=1, stride=2), (2, 8, 16)),
    ("convT_k4s2_32_16", "transp conv", dict(in_channels=32, out_channels=16, kernel_size=4, padding=1, stride=2), (2, 16, 16)),
    # CGAN vocabulary (trained_models/README.md:106-128): bias, k9, k3s2, output_padding
    ("conv_k9s1_2_32_bias", "conv", dict(in_channels=2, out_channels=32, kernel_size=9, padding=4, stride=1, bias=True), (2, 16, 16)),
    ("conv_k3s2_32_64_bias", "conv", dict(in_channels=32, out_channels=64, kernel_size=3, padding=1, stride=2, bias=True), (2, 16, 16)),
    ("convT_k3s2op1_64_32_bias", "transp conv", dict(in_channels=64, out_channels=32, kernel_size=3, padding=1, stride=2, output_padding=1, bias=True), (2, 8, 8)),
]


