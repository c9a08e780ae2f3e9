"""`from data.dataset import BaseDataset` of the reference trainer -> vq_seg_amd.data (see compat/_vqseg_compat.py)."""
from _vqseg_compat import bind as _bind

_bind(__name__, "vq_seg_amd.data")
