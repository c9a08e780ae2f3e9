"""`from measurement import Measurement` of the reference trainer -> vq_seg_amd.measurement (see compat/_vqseg_compat.py)."""
from _vqseg_compat import bind as _bind

_bind(__name__, "vq_seg_amd.measurement")
