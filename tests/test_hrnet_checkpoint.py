"""HRNet-W48 checkpoints: the module must carry the official `pose_hrnet_w48_384x288.pth` key layout (the
reference names the file in /root/reference/src/configs/Shelf/model_configs.yaml:49-58) so that the authors' weights load when they
are available; offline only a synthetic state dict in that layout can be round-tripped."""
import os

import pytest
import torch

from pam import hrnet

OFFICIAL_KEYS = [
    'conv1.weight', 'bn1.running_mean', 'conv2.weight', 'bn2.weight',
    'layer1.0.conv1.weight', 'layer1.0.bn3.running_var', 'layer1.0.downsample.0.weight', 'layer1.0.downsample.1.bias', 'layer1.3.conv3.weight',
    'transition1.0.0.weight', 'transition1.0.1.weight', 'transition1.1.0.0.weight', 'transition1.1.0.1.running_mean',
    'stage2.0.branches.0.0.conv1.weight', 'stage2.0.branches.1.3.bn2.running_var',
    'stage2.0.fuse_layers.0.1.0.weight', 'stage2.0.fuse_layers.0.1.1.weight', 'stage2.0.fuse_layers.1.0.0.0.weight', 'stage2.0.fuse_layers.1.0.0.1.bias',
    'transition2.2.0.0.weight', 'transition2.2.0.1.weight',
    'stage3.3.branches.2.3.conv2.weight', 'stage3.0.fuse_layers.2.0.0.0.weight', 'stage3.0.fuse_layers.2.0.1.0.weight', 'stage3.0.fuse_layers.2.0.1.1.weight',
    'transition3.3.0.0.weight',
    'stage4.2.branches.3.3.bn2.weight', 'stage4.0.fuse_layers.3.0.2.0.weight', 'stage4.0.fuse_layers.0.3.0.weight', 'stage4.2.fuse_layers.0.1.1.running_var',
    'final_layer.weight', 'final_layer.bias',
]


def test_state_dict_has_the_official_key_layout():
    m = hrnet.PoseHighResolutionNet(48, 17)
    sd = m.state_dict()
    for k in OFFICIAL_KEYS:
        assert k in sd, k
    assert sd['conv1.weight'].shape == (64, 3, 3, 3) and sd['layer1.0.downsample.0.weight'].shape == (256, 64, 1, 1)
    assert sd['stage4.0.fuse_layers.3.0.2.0.weight'].shape == (384, 48, 3, 3) and sd['stage4.0.fuse_layers.0.3.0.weight'].shape == (48, 384, 1, 1)
    assert sd['final_layer.weight'].shape == (17, 48, 1, 1)
    # the last stage-4 module fuses to branch 0 only (multi_scale_output=False in the official network)
    assert not any(k.startswith('stage4.2.fuse_layers.1.') for k in sd)
    n_conv = sum(v.numel() for k, v in sd.items() if v.dim() == 4)
    assert abs(n_conv - 63.5e6) < 0.2e6, n_conv


@pytest.mark.gpu
def test_checkpoint_file_round_trip(tmp_path):
    """A state dict saved in the official layout (also wrapped as {'model': ...}) loads through CHECKPOINT_FILE exactly as the YAML
    drives it, and the product path then computes with THOSE weights (same heat-maps as a module initialised from the same tensors)."""
    src = hrnet.init_random(hrnet.PoseHighResolutionNet(48, 17), seed=5)
    b = hrnet.HRNetPose(48, 17, None, resolution=(384, 288), use_graph=False, seed=5)       # the same tensors, never through a file
    c = hrnet.HRNetPose(48, 17, None, resolution=(384, 288), use_graph=False, seed=6)       # other weights
    x = b.input_buffer(1)
    x.copy_(torch.randn(x.shape, device=x.device).to(x.dtype)); x[:, 3:] = 0
    yb, yc = b.heatmaps(x).clone(), c.heatmaps(x).clone()
    for wrap in (False, True):
        path = os.path.join(str(tmp_path), 'pose_hrnet_w48_384x288%s.pth' % ('_w' if wrap else ''))
        torch.save({'model': src.state_dict()} if wrap else src.state_dict(), path)
        a = hrnet.HRNetPose(48, 17, path, resolution=(384, 288), use_graph=False)
        assert a.weights == path
        ya = a.heatmaps(x)
        torch.cuda.synchronize()
        assert torch.equal(ya, yb)
        assert not torch.equal(ya, yc)
