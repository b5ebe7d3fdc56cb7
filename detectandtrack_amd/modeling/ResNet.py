"""2D ResNet bodies — same graph as ResNet3D with T = 1 and kT = 1 (reference lib/modeling/ResNet.py:231-397).
Blob and parameter names are identical to the 2D reference (conv1, res_conv1_bn, res2_0_branch2a, ...), and 2D
checkpoints load into either module (4-D weights are inflated to [.., 1, k, k], utils/net.py)."""
from detectandtrack_amd.core.config import cfg
import detectandtrack_amd.modeling.ResNet3D as R3


def _as_2d(fn):
    def wrapped(model, *a, **k):
        saved = cfg.VIDEO.TIME_KERNEL_DIM.BODY
        cfg.VIDEO.TIME_KERNEL_DIM.BODY = 1
        try:
            return fn(model, *a, **k)
        finally:
            cfg.VIDEO.TIME_KERNEL_DIM.BODY = saved
    wrapped.__name__ = fn.__name__
    return wrapped


add_ResNet18_conv4_body = _as_2d(R3.add_ResNet18_conv4_body)
add_ResNet18_conv5_body = _as_2d(R3.add_ResNet18_conv5_body)
add_ResNet34_conv4_body = _as_2d(R3.add_ResNet34_conv4_body)
add_ResNet34_conv5_body = _as_2d(R3.add_ResNet34_conv5_body)
add_ResNet50_conv4_body = _as_2d(R3.add_ResNet50_conv4_body)
add_ResNet50_conv5_body = _as_2d(R3.add_ResNet50_conv5_body)
add_ResNet101_conv4_body = _as_2d(R3.add_ResNet101_conv4_body)
add_ResNet101_conv5_body = _as_2d(R3.add_ResNet101_conv5_body)
add_ResNet152_conv5_body = _as_2d(R3.add_ResNet152_conv5_body)
add_ResNet18_roi_conv5_head = R3.add_ResNet18_roi_conv5_head
add_ResNet34_roi_conv5_head = R3.add_ResNet34_roi_conv5_head
add_ResNet_roi_conv5_head = R3.add_ResNet_roi_conv5_head
add_stage = R3.add_stage
stage_info_ResNet18_conv5 = R3.stage_info_ResNet18_conv5
stage_info_ResNet34_conv5 = R3.stage_info_ResNet34_conv5
stage_info_ResNet50_conv5 = R3.stage_info_ResNet50_conv5
stage_info_ResNet101_conv5 = R3.stage_info_ResNet101_conv5
stage_info_ResNet152_conv5 = R3.stage_info_ResNet152_conv5
