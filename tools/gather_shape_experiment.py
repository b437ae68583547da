"""What a wider node would cost the vector-memory path (GPU box, with a library variant whose ceiling kernel takes the node
size and the number of 16-byte loads per node from STHIP_CEIL_NODE_BYTES / STHIP_CEIL_LOADS): independent random node
fetches, as sthip_measure_ceiling does them, for several node shapes; G nodes/s is what a traversal step costs.
usage: tools/build_variant.sh ceil tools/variant_patches/ceiling_shapes.py && STHIP_LIB=$PWD/_variants/ceil.so python tools/gather_shape_experiment.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratum_amd import scenes
from stratum_amd.bdpt import BDPT

sc, cam = scenes.SCENES["atrium"]()
r = BDPT(0)
r.update(sc)
for nb, loads in ((48, 3), (64, 3), (64, 4), (32, 2), (80, 5), (128, 4), (128, 7), (128, 8)):
    os.environ["STHIP_CEIL_NODE_BYTES"] = str(nb)
    os.environ["STHIP_CEIL_LOADS"] = str(loads)
    row = []
    for kind in ("node_gather_l1", "node_gather_l2", "node_gather_table"):
        g = r.measure_ceiling(kind)
        row.append("%s %7.0f GB/s %6.1f Gnode/s" % (kind[12:], g, g / (16.0 * loads)))
    print("stride %3d B, %d loads: %s" % (nb, loads, " | ".join(row)), flush=True)
r.close()
