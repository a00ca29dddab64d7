"""Constants the decode path reads.

The reference keeps these in an EasyDict built at import time
(cfgs/config_vgg19.py:20-21,48,56-63,65,73-77,80-81; cfgs/config_mobilenetv2.py
holds the same values, only ``backbone`` differs).  Here they are a plain
immutable table so the decode has no import-time side effects.
"""

# cfgs/config_vgg19.py:56-57
LIMB_FROM = (1, 8, 9, 1, 11, 12, 1, 2, 3, 2, 1, 5, 6, 5, 1, 0, 0, 14, 15)
LIMB_TO = (8, 9, 10, 11, 12, 13, 2, 3, 4, 16, 5, 6, 7, 17, 0, 14, 15, 16, 17)

# cfgs/config_vgg19.py:59-63 -- lists of lists, like the reference's cfg.limb_seq
limb_seq = [[a, b] for a, b in zip(LIMB_FROM, LIMB_TO)]
map_idx = [[2 * k, 2 * k + 1] for k in range(len(LIMB_FROM))]

N_PARTS = 18            # cfg.ch_heats - 1 (config_vgg19.py:76); channel 18 is background
N_LIMBS = 19            # len(cfg.map_idx)
CH_HEATS = 19           # config_vgg19.py:76
CH_VECTORS = 38         # config_vgg19.py:77
STRIDE = 8              # config_vgg19.py:65
IMG_Y = 368             # config_vgg19.py:20
IMG_X = 368             # config_vgg19.py:21
THRE1 = 0.1             # config_vgg19.py:73
THRE2 = 0.05            # config_vgg19.py:74
SCALE_SEARCH = (0.5, 1, 1.5, 2)   # config_vgg19.py:80
PAD_VALUE = 128         # config_vgg19.py:81
MID_NUM = 10            # literal in utils/postprocess.py:79 and predict.py:123
GAUSS_SIGMA = 3         # utils/postprocess.py:44, predict.py:93
GAUSS_RADIUS = 12       # scipy: int(truncate * sigma + 0.5) with truncate = 4.0

# cfgs/config_vgg19.py:48 -- COCO keypoint index -> index in the 18-part order
coco_to_ours = [0, 15, 14, 17, 16, 5, 2, 6, 3, 7, 4, 11, 8, 12, 9, 13, 10]

MODE_P = 0   # utils/postprocess.py semantics
MODE_D = 1   # predict.py detect() semantics
