"""`afm(lines, shape_info, height, width)` — drop-in for the reference's CUDA extension op
(pixelspointspolygons/models/hisup/afm_module/afm_op: `afm_cuda`, called at models/hisup/model_hisup.py:95) on the HIP kernel
`p3_afm`.  lines [L,4] float (x1, y1, x2, y2), shape_info int [B,4] = (start, end, src_height, src_width) on the device ->
(afmap float32 [B,2,height,width], aflabel int32 [B,1,height,width]).  No CPU fallback: raises without libp3hip.so / a GPU tensor."""
from . import hip


def afm(lines, shape_info, height, width):
    return hip.afm(lines, shape_info, int(height), int(width))
