"""ConvStageInfo (reference lib/modeling/common.py)."""
import collections

ConvStageInfo = collections.namedtuple('ConvStageInfo', ['blobs', 'dims', 'spatial_scales'])
