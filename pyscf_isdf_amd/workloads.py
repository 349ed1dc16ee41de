"""The BASELINE.json workloads as concrete inputs (cells, meshes, density matrices)."""
import numpy as np
from . import gto

WORKLOADS = {
    # name: (builder, description)
    'diamond-prim-szv-40': (lambda: gto.diamond_primitive('gth-szv', (40, 40, 40)),
                            'configs[0]: diamond primitive cell, gth-szv, 40^3'),
    'diamond-222-dzvp-80': (lambda: gto.diamond_supercell(2, 'gth-dzvp', (80, 80, 80)),
                            'configs[1]: diamond 2x2x2, gth-dzvp, 80^3, c_isdf=10'),
    'diamond-444-dzvp-120': (lambda: gto.diamond_supercell(4, 'gth-dzvp', (120, 120, 120)),
                             'configs[2]: diamond 4x4x4, gth-dzvp, 120^3, c_isdf=10'),
    'diamond-333-dzvp-96': (lambda: gto.diamond_supercell(3, 'gth-dzvp', (96, 96, 96)),
                            'intermediate: diamond 3x3x3, gth-dzvp, 96^3'),
    'mgo-333-dzvp-k222': (lambda: mgo_supercell(3, 'gth-dzvp', (96, 96, 96)),
                          'configs[3]: MgO 3x3x3, gth-dzvp, 96^3, 2x2x2 k-mesh'),
    'mgo-222-dzvp-k222': (lambda: mgo_supercell(2, 'gth-dzvp', (64, 64, 64)),
                          'reduced configs[3]: MgO 2x2x2, gth-dzvp, 64^3, 2x2x2 k-mesh'),
}
KMESH = {'mgo-333-dzvp-k222': [2, 2, 2], 'mgo-222-dzvp-k222': [2, 2, 2]}


def mgo_supercell(n, basis, mesh):
    """MgO rocksalt, a = 4.213 A (pyscf/pbc/tools/lattice.py:124), n x n x n of the fcc primitive cell."""
    a0 = 4.213
    a = np.array([[0., a0 / 2, a0 / 2], [a0 / 2, 0., a0 / 2], [a0 / 2, a0 / 2, 0.]])
    prim = gto.Cell(atom=[('Mg', (0., 0., 0.)), ('O', (a0 / 2, a0 / 2, a0 / 2))], a=a, basis=basis, mesh=(8, 8, 8),
                    pseudo='gth-pade')
    return gto.super_cell(prim, [n, n, n], mesh=mesh)


def make_kpts(name, cell):
    return cell.make_kpts(KMESH[name]) if name in KMESH else None


def make_cell(name):
    return WORKLOADS[name][0]()


def make_dm(cell, seed=20240203):
    """D = C diag(occ) C^T, C = qr(N(0,1))[0], occ = 2 on nelec/2 columns."""
    nao = cell.nao_nr()
    rng = np.random.default_rng(seed)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao)
    occ[:cell.nelectron // 2] = 2
    return (c * occ).dot(c.T), c, occ
