"""Host-side helper of the k-point path: the set of distinct difference vectors q = k2 - k1 of a k-point list.
(The Coulomb kernel table for a q lives on the device: include/mi355_isdf.h isdf_coulG_q.)"""
import numpy as np


def unique_q(kpts, kpts_band=None, tol=1e-9):
    """Distinct q = k2 - k1 (k1 in band, k2 in kpts) and index[k1][k2] -> position in the q list."""
    kpts = np.reshape(kpts, (-1, 3))
    band = kpts if kpts_band is None else np.reshape(kpts_band, (-1, 3))
    qs, index = [], np.zeros((len(band), len(kpts)), dtype=int)
    for i1, k1 in enumerate(band):
        for i2, k2 in enumerate(kpts):
            q = k2 - k1
            for iq, qq in enumerate(qs):
                if abs(qq - q).max() < tol:
                    index[i1, i2] = iq
                    break
            else:
                qs.append(q)
                index[i1, i2] = len(qs) - 1
    return np.array(qs), index
