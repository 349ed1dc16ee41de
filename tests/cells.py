"""Test cells: the reference's own fixtures restated as inputs (data, not code)."""
import numpy as np
from pyscf_isdf_amd import gto


def cell_he_c():
    """pyscf/pbc/df/test/test_fft.py:545-555: He + C(gth-szv) in a 2.5 A cube, mesh 21^3."""
    return gto.Cell(atom='He 1. .5 .5; C .1 1.3 2.1',
                    basis={'He': [(0, (2.5, 1)), (0, (1., 1))], 'C': 'gth-szv'},
                    pseudo={'C': 'gth-pade'}, a=np.eye(3) * 2.5, mesh=[21] * 3)


def cell_he2_triclinic():
    """pyscf/pbc/df/test/test_fft.py:566-577: triclinic He2 cell with s and p shells, mesh 17^3."""
    return gto.Cell(atom='He 1.3 .2 .3; He .1 .1 1.1',
                    basis={'He': [[0, [0.8, 1]], [1, [0.6, 1]]]},
                    a=np.array(([2.0, .9, 0.], [0.1, 1.9, 0.4], [0.8, 0, 2.1])), mesh=[17] * 3)


# C cc-pVDZ shell table (pyscf/gto/basis/cc-pvdz.dat:96-114), used by test_numint.py:77-96
CCPVDZ_C = [
    [0, [6665.0, 0.000692, -0.000146], [1000.0, 0.005329, -0.001154], [228.0, 0.027077, -0.005725],
        [64.71, 0.101718, -0.023312], [21.06, 0.27474, -0.063955], [7.495, 0.448564, -0.149981],
        [2.797, 0.285074, -0.127262], [0.5215, 0.015204, 0.544529]],
    [0, [0.1596, 1.0]],
    [1, [9.439, 0.038109], [2.002, 0.20948], [0.5456, 0.508557]],
    [1, [0.1517, 1.0]],
    [2, [0.55, 1.0]],
]


def cell_c2_ccpvdz():
    """pyscf/pbc/dft/test/test_numint.py:78-87."""
    return gto.Cell(atom=[['C', (1., .8, 1.9)], ['C', (.1, .2, .3)]], basis={'C': CCPVDZ_C},
                    a=np.eye(3) * 2.5, mesh=[21] * 3, precision=1e-11)


def cell_diamond_prim(basis='gth-szv', mesh=(12, 12, 12)):
    return gto.diamond_primitive(basis=basis, mesh=mesh)
