"""Common base of the four networks (reference: models/networks/base_network.py:4-57)."""
import torch


class BaseNetwork(torch.nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt

    def print_architecture(self, verbose=False):
        lines = ["------------------- %s -------------------" % type(self).__name__]
        total = 0
        for name, child in self.named_children():
            n = sum(p.numel() for p in child.parameters())
            total += n
            if verbose:
                lines.append("%s: %.3fM" % (name, n / 1e6))
        lines.append("[Network %s] Total number of parameters : %.3f M" % (type(self).__name__, total / 1e6))
        print("\n".join(lines))

    def set_requires_grad(self, requires_grad):
        for p in self.parameters():
            p.requires_grad = requires_grad

    def collect_parameters(self, name):
        return [p for m in self.modules() if type(m).__name__ == name for p in m.parameters()]

    def fix_and_gather_noise_parameters(self):
        device = next(self.parameters()).device
        params = []
        for m in self.modules():
            if type(m).__name__ == "NoiseInjection":
                assert m.image_size is not None, "run one forward pass before fixing the noise"
                b, _, h, w = m.image_size
                m.fixed_noise = torch.nn.Parameter(torch.randn(b, 1, h, w, device=device))
                params.append(m.fixed_noise)
        return params

    def remove_noise_parameters(self, name=None):
        for m in self.modules():
            if type(m).__name__ == "NoiseInjection":
                m.fixed_noise = None

    def forward(self, x):
        return x
