"""PyTorch-ROCm forms of the product's networks, for the GPU tests to compare against (test infrastructure: nothing under the package
imports this).  Until round 5 the bf16 PyTorch conv path lived inside the product class as ``HRNetPose(backend='miopen')``."""
import torch


def bf16_torch_heatmaps(x, seed=0, device=None):
    """Heat-maps of the folded HRNet-W48 (random weights of `seed`, as HRNetPose builds them) run by PyTorch-ROCm's own bf16
    convolutions (MIOpen), channels-last: x (N, 3, H, W) bf16 -> (N, 17, H/4, W/4) float32.  The 1x1 head runs in float32, as in the product."""
    from pam import hrnet
    device = device or x.device
    model = hrnet.fold_batchnorm(hrnet.init_random(hrnet.PoseHighResolutionNet(), seed=seed))
    head = model.final_layer.to(device).float()
    model.final_layer = torch.nn.Identity()
    model = model.to(device).to(torch.bfloat16).to(memory_format=torch.channels_last).eval()
    with torch.no_grad():
        f = model.features(x.to(device).contiguous(memory_format=torch.channels_last))
        return head(f.float())
