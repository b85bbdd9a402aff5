import os, sys
sys.path.insert(0, "tests")
import numpy as np
import golden_util as gu
from cuclark_amd import MiClarkDB
rng = np.random.default_rng(23)
k, T, htsize = 31, 9, 2000003
sizes, keys, labels, canon = gu.random_db(rng, htsize, 150000, k, 8, T)
os.environ["MIC_LAYOUT"] = "minimizer"
for load in (4, 5, 6, 7, 8, 9, 10, 12):
    os.environ["MIC_MSLOT_LOAD"] = str(load)
    with MiClarkDB(k, T) as e:
        e.read_arrays(sizes, keys, labels)
        i = e.info()
        print(load, i["n_slots"], i["n_overflow"], i["hbm_bytes"], i["max_chain"])
