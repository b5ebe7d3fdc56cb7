#!/usr/bin/env python3
"""Tracking entry point — mirror of reference tools/compute_tracks.py:43-56: loads <output_dir>/detections.pkl (or
TRACKING.DETECTIONS_FILE), runs the host Hungarian tracker, writes detections_withTracks.pkl."""
import argparse
import logging
import pickle

import _path  # noqa
from detectandtrack_amd.core.config import cfg_from_file, cfg_from_list, assert_and_infer_cfg, get_output_dir
from detectandtrack_amd.core.tracking_engine import run_posetrack_tracking


def main():
    logging.basicConfig(level=logging.INFO)
    p = argparse.ArgumentParser()
    p.add_argument('--cfg', dest='cfg_file', required=True)
    p.add_argument('--roidb', default='', help='pickled clip list (needs image/height/width per entry); default: cfg.TEST.DATASET with its ground truth, '
                                               'like the reference (tools/compute_tracks.py:43-56)')
    p.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    args = p.parse_args()
    cfg_from_file(args.cfg_file)
    if args.opts:
        cfg_from_list(args.opts)
    assert_and_infer_cfg()
    if args.roidb:
        with open(args.roidb, 'rb') as f:
            roidb = pickle.load(f)
        json_data = [{'image': e.get('name', 'images/vid0000/%06d.jpg' % i), 'height': e['height'], 'width': e['width']}
                     for i, e in enumerate(roidb)]
    else:
        from detectandtrack_amd.core.test_engine import get_roidb_and_dataset
        from detectandtrack_amd.datasets.json_dataset import load_catalog_from_env
        load_catalog_from_env()
        json_data = get_roidb_and_dataset(None, include_gt=True)[0]
    run_posetrack_tracking(get_output_dir(training=False), json_data)


if __name__ == '__main__':
    main()
