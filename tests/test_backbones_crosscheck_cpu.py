"""CPU: the restated EfficientNet-B4 trunk against an INDEPENDENT implementation of the same published network.

The reference's trunk is ``efficientnet_pytorch==0.7.0`` (stp3/models/encoder.py:8, :21-37), whose source is not under
/root/reference, so stp3_amd/models/efficientnet.py restates the published architecture ("parity unpinned",
DESIGN.md section 2).  Hugging Face ``transformers`` (installed here, unrelated code base) ships its own
EfficientNet; with the B4 coefficients (width 1.4, depth 1.8) and the B4 padding quirk of the converted checkpoints
(``depthwise_padding=[6]``: the first 5x5 stride-2 block pads symmetrically, because the 380-pixel chain reaches it at
an odd size -- the same thing efficientnet_pytorch's *static* same-padding bakes in) it has the same parameter
tensors in the same order.  Copying OUR weights into it position by position and comparing every block output on a
224x480 input checks block structure, strides, expansion / squeeze ratios, padding and BatchNorm epsilon of the
restatement against a second implementation (eval mode, float32, rtol 1e-4)."""
import pytest
import torch

transformers = pytest.importorskip('transformers')


def test_trunk_matches_the_transformers_implementation_block_by_block():
    from transformers import EfficientNetConfig, EfficientNetModel
    from stp3_amd.models.efficientnet import EfficientNet

    torch.manual_seed(0)
    mine = EfficientNet('efficientnet-b4').eval()
    with torch.no_grad():                                  # non-trivial statistics and affine parameters
        for m in mine.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    cfg = EfficientNetConfig(width_coefficient=1.4, depth_coefficient=1.8, image_size=380, depthwise_padding=[6],
                             hidden_dim=1792)
    other = EfficientNetModel(cfg).eval()
    ours = [(k, v) for k, v in mine.state_dict().items() if not k.startswith('_fc.')]
    theirs = other.state_dict()
    assert len(ours) == len(theirs) == 704
    mapped = {}
    for (ko, vo), (kt, vt) in zip(ours, theirs.items()):
        assert vo.shape == vt.shape, (ko, kt, vo.shape, vt.shape)
        mapped[kt] = vo.clone()
    other.load_state_dict(mapped)
    assert len(other.encoder.blocks) == len(mine._blocks) == 32

    x = torch.randn(2, 3, 224, 480)
    with torch.no_grad():
        ref = other(pixel_values=x, output_hidden_states=True).hidden_states      # embeddings, then every block
        y = mine._swish(mine._bn0(mine._conv_stem(x)))
        torch.testing.assert_close(y, ref[0], rtol=1e-4, atol=1e-4)
        for i, blk in enumerate(mine._blocks):
            y = blk(y)
            assert y.shape == ref[i + 1].shape, (i, y.shape, ref[i + 1].shape)
            torch.testing.assert_close(y, ref[i + 1], rtol=1e-4, atol=1e-4, msg=lambda m, i=i: f'block {i}: {m}')
    # the two endpoints the ST-P3 encoder taps (encoder.py:21: reduction_3 = 56 channels at /8, reduction_4 = 160 at /16)
    assert ref[10].shape == (2, 56, 28, 60) and ref[22].shape == (2, 160, 14, 30)


def test_resnet18_stages_match_the_transformers_implementation():
    """Same idea for the BEV decoder's backbone: ``torchvision==0.11.3`` resnet18 ``layer1-3`` (stp3/models/decoder.py:
    22-30) restated in stp3_amd/models/resnet.py, against transformers' ResNet with basic layers (depths 2-2-2-2)."""
    from transformers import ResNetConfig, ResNetModel
    from stp3_amd.models import resnet

    torch.manual_seed(1)
    mine = resnet.resnet18(zero_init_residual=False).eval()
    with torch.no_grad():
        for m in mine.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2],
                       layer_type='basic', hidden_act='relu', downsample_in_first_stage=False)
    other = ResNetModel(cfg).eval()
    sd = other.state_dict()
    used = 0
    for k, v in mine.state_dict().items():
        if not k.startswith('layer'):
            continue
        stage, block, rest = k.split('.', 2)
        base = f'encoder.stages.{int(stage[5:]) - 1}.layers.{block}.'
        part, leaf = rest.split('.', 1) if not rest.startswith('downsample') else (rest[:12], rest[13:])
        target = {'conv1': 'layer.0.convolution.', 'bn1': 'layer.0.normalization.', 'conv2': 'layer.1.convolution.',
                  'bn2': 'layer.1.normalization.', 'downsample.0': 'shortcut.convolution.',
                  'downsample.1': 'shortcut.normalization.'}[part]
        key = base + target + leaf
        assert sd[key].shape == v.shape, (k, key)
        sd[key] = v.clone()
        used += 1
    assert used == 84
    other.load_state_dict(sd)
    x = torch.randn(2, 64, 100, 100)
    with torch.no_grad():
        y, z = x, x
        for name, stage in (('layer1', other.encoder.stages[0]), ('layer2', other.encoder.stages[1]),
                            ('layer3', other.encoder.stages[2])):
            y = getattr(mine, name)(y)
            z = stage(z)
            torch.testing.assert_close(y, z, rtol=1e-4, atol=1e-4, msg=lambda m, name=name: f'{name}: {m}')
    assert y.shape == (2, 256, 25, 25)
