import sys; sys.path.insert(0,'.')
import numpy as np, lz77_amd as L
from lz77_amd import synth
d=synth.text(100_000_000, 0x5EED0001)
P,S=L.stage_neighbours(d)
(P.astype(np.uint32)|(S.astype(np.uint32)<<16)).tofile('/dev/shm/ps.bin')
print("ok")
