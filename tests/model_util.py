"""Shared helpers for the end-to-end model tests / smoke(): build the product model + the oracle on the same
synthetic weights and inputs."""
import numpy as np


def fpn3d_kps_cfg(arch='18', T=4, kt=3, link='slice-center', pre=300, post=100, dtype='fp32'):
    return {
        'MODEL': {'TYPE': 'keypoint_rcnn', 'CONV_BODY': 'FPN3D.add_fpn_ResNet%s_conv5_body' % arch,
                  'ROI_HEAD': 'head_builder.add_roi_2mlp_head', 'NUM_CLASSES': 2, 'FASTER_RCNN': True,
                  'KEYPOINTS_ON': True, 'VIDEO_ON': True},
        'FPN': {'FPN_ON': True, 'MULTILEVEL_ROIS': True, 'MULTILEVEL_RPN': True},
        'FAST_RCNN': {'ROI_XFORM_METHOD': 'RoIAlign', 'ROI_XFORM_RESOLUTION': 7, 'ROI_XFORM_SAMPLING_RATIO': 2},
        'KRCNN': {'ROI_KEYPOINTS_HEAD': 'keypoint_rcnn_heads.add_roi_pose_head_v1convX', 'NUM_STACKED_CONVS': 8,
                  'NUM_KEYPOINTS': 17, 'USE_DECONV_OUTPUT': True, 'CONV_INIT': 'MSRAFill', 'CONV_HEAD_DIM': 512,
                  'UP_SCALE': 2, 'HEATMAP_SIZE': 56, 'ROI_XFORM_METHOD': 'RoIAlign', 'ROI_XFORM_RESOLUTION': 14,
                  'ROI_XFORM_SAMPLING_RATIO': 2},
        'VIDEO': {'NUM_FRAMES': T, 'TIME_KERNEL_DIM': kt, 'BODY_HEAD_LINK': link, 'WEIGHTS_INFLATE_MODE': 'center-only'},
        'TEST': {'RPN_PRE_NMS_TOP_N': pre, 'RPN_POST_NMS_TOP_N': post, 'COMPETITION_MODE': False, 'NMS': 0.5},
        'HIP': {'DTYPE': dtype},
    }


def build_product(cfg_dict, seed=3):
    """Returns (model, workspace, weights dict).  Needs the GPU."""
    from detectandtrack_amd.core.config import cfg, cfg_from_cfg, assert_and_infer_cfg, reset_cfg
    from detectandtrack_amd.modeling import model_builder
    from detectandtrack_amd.utils import net as net_utils
    from detectandtrack_amd import workspace
    reset_cfg()
    cfg_from_cfg(cfg_dict)
    assert_and_infer_cfg()
    model = model_builder.create(cfg.MODEL.TYPE, train=False)
    workspace.ResetWorkspace()
    ws = workspace.GlobalWorkspace()
    weights = net_utils.synthetic_params(model, seed)
    for k, v in weights.items():
        ws.set_param(k, v)
    ws.CreateNet(model.net)
    ws.CreateNet(model.conv_body_net)
    if model.keypoint_net is not None:
        ws.CreateNet(model.keypoint_net)
    return model, ws, weights


def synthetic_clip(T, H, W, seed=3):
    """uint8-like BGR frames minus PIXEL_MEANS, NC(T)HW fp32."""
    rs = np.random.RandomState(seed)
    means = np.array([102.9801, 115.9465, 122.7717], dtype=np.float32).reshape(1, 3, 1, 1, 1)
    # smooth-ish content so activations are not pure noise
    base = rs.uniform(0, 255, (1, 3, T, H // 8 + 1, W // 8 + 1)).astype(np.float32)
    data = np.repeat(np.repeat(base, 8, axis=3), 8, axis=4)[:, :, :, :H, :W]
    data = data + rs.uniform(-20, 20, data.shape).astype(np.float32)
    return np.clip(data, 0, 255) - means


def oracle_opts(arch, T, kt, link, pre, post):
    from oracle.net3d import opts_for
    return opts_for('R' + arch, kt_body=kt, body_head_link=link, num_frames_mid=T, pre_nms_topn=pre,
                    post_nms_topn=post)


def c4_tube_kps_cfg(T=3, kt=3, pre=300, post=60, dtype='fp32', deconv='time_to_batch'):
    """The shipped 3D configuration (configs/video/3d/04_R-18-3D_PTFromImNet.yaml): ResNet-18 3D C4 body, tube RPN,
    per-RoI res5 head, 3D keypoint head.  deconv: 'time_to_batch' (KRCNN.NO_3D_DECONV_TIME_TO_CH True, what the shipped configs
    set) | 'grouped' (the reference default: time -> channels + ConvTranspose group = T) | 'group_ignored' (the same graph with the
    full brew filter and the group argument dropped, cfg.HIP.DECONV_GROUP_IGNORED)."""
    d = fpn3d_kps_cfg('18', T=T, kt=kt, link='', pre=pre, post=post, dtype=dtype)
    d['MODEL'].update(CONV_BODY='ResNet3D.add_ResNet18_conv4_body', ROI_HEAD='ResNet3D.add_ResNet18_roi_conv5_head')
    d['FPN'] = {'FPN_ON': False, 'MULTILEVEL_ROIS': False, 'MULTILEVEL_RPN': False}
    assert deconv in ('time_to_batch', 'grouped', 'group_ignored')
    d['KRCNN'].update(ROI_KEYPOINTS_HEAD='keypoint_rcnn_heads.add_roi_pose_head_v1convX_3d',
                      NO_3D_DECONV_TIME_TO_CH=(deconv == 'time_to_batch'))
    d['HIP']['DECONV_GROUP_IGNORED'] = (deconv == 'group_ignored')
    return d


def fpn3d_tube_kps_cfg(T=2, kt=3, pre=200, post=50, dtype='fp32'):
    """Declared extension: FPN3D body kept 3D (BODY_HEAD_LINK ''), tube RPN per level, tube rois
    on the 2-MLP box head, 3D keypoint head."""
    d = fpn3d_kps_cfg('18', T=T, kt=kt, link='', pre=pre, post=post, dtype=dtype)
    d['KRCNN'].update(ROI_KEYPOINTS_HEAD='keypoint_rcnn_heads.add_roi_pose_head_v1convX_3d',
                      NO_3D_DECONV_TIME_TO_CH=True)
    return d


def fpn2d_kps_cfg(arch='50', pre=1000, post=1000, dtype='fp32'):
    """BASELINE configs 1-2: the pure 2D R-50-FPN keypoint R-CNN (MODEL.VIDEO_ON False; reference lib/modeling/FPN.py:114-202,
    ResNet.py:231-266; shipped as configs/video/2d_best/01_R101_best_hungarian.yaml with the R-101 body)."""
    d = fpn3d_kps_cfg(arch, T=1, kt=1, link='', pre=pre, post=post, dtype=dtype)
    d['MODEL'].update(CONV_BODY='FPN.add_fpn_ResNet%s_conv5_body' % arch, VIDEO_ON=False)
    d.pop('VIDEO')
    return d


def oracle_weights_2d(weights):
    """The oracle graph is written on 5-D blobs: 2D conv weights of the body / FPN ([o, i, k, k]) get a unit time axis."""
    out = {}
    for k, v in weights.items():
        v = np.asarray(v)
        if v.ndim == 4 and k.endswith('_w') and k.startswith(('conv1', 'res', 'fpn_')):
            v = v[:, :, None]
        out[k] = v
    return out
