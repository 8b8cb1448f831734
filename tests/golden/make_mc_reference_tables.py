#!/usr/bin/env python3
"""Makes tests/golden/mc_reference_tables.npz: the two constant lookup tables of the reference's marching cubes
(utopian/shaders/marching_cubes/tables.glsl:4-293: edgeTable[256], triangleTable[256][16]) as plain integer arrays - the
data the oracle's restatement of marching_cubes.comp:179-254 (oracle/oracle.cpp orc_marching_cubes) runs on. Numbers only; run
where the reference checkout is mounted."""
import os
import re

import numpy as np

SRC = "/root/reference/utopian/shaders/marching_cubes/tables.glsl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mc_reference_tables.npz")

text = re.sub(r"/\*.*?\*/|//[^\n]*", "", open(SRC).read(), flags=re.S)
edge_body = text[text.index("edgeTable"):text.index("triangleTable")]
edge = np.array([int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]+", edge_body[edge_body.index("{"):])], dtype=np.int32)
tri_body = text[text.index("triangleTable"):]
tri = np.array([int(t) for t in re.findall(r"-?\d+", tri_body[tri_body.index("{"):])], dtype=np.int32)
assert edge.shape == (256,) and tri.size == 256 * 16, (edge.shape, tri.size)
np.savez_compressed(OUT, edge_table=edge, triangle_table=tri.reshape(256, 16))
print("wrote", OUT, os.path.getsize(OUT), "bytes")
