"""hcore pieces of ``isdf.ISDF``: ``get_nuc`` / ``get_pp`` as FFTDF has them (pyscf/pbc/df/fft.py:39-152; DESIGN.md
section 6c).  Host orchestration only."""
import numpy as np
import torch
from . import gto


class HcoreMixin:
    def get_pp(self, kpts=None):
        """GTH pseudopotential AO matrix (G=0 removed), pyscf/pbc/df/fft.py:64-152: local part on the FFT
        grid, non-local part from projector/AO overlaps in reciprocal space — both on the device
        (pp.hip); only the final nproj-sized contraction with the h_ij matrices runs on the host.
        Returns (nao,nao) for Gamma / a single k-point, else (nk,nao,nao)."""
        cell, be = self.cell, self.backend
        pseudo = getattr(cell, '_pseudo', None) or {}
        if kpts is None:
            kpts_lst, single = np.zeros((1, 3)), True
        else:
            kpts_lst = np.reshape(kpts, (-1, 3))
            single = np.ndim(kpts) == 1
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        nao = cell.nao_nr()
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        acoords = np.asarray(cell.atom_coords(), dtype=float)
        charges = np.asarray(cell.atom_charges(), dtype=float)
        pp_par = np.zeros((cell.natm, 8))
        proj_tab, proj_rl, blocks = [], [], []          # blocks: (row0, l, nl, h) per atom and l-channel
        row = 0
        for ia in range(cell.natm):
            pp = pseudo.get(cell.atom_symbol(ia))
            pp_par[ia, 1] = charges[ia]
            if pp is None:
                continue
            rloc, nexp, cexp = pp[1], pp[2], pp[3]
            pp_par[ia, 0], pp_par[ia, 2], pp_par[ia, 3] = 1.0, rloc, nexp
            pp_par[ia, 4:4 + nexp] = cexp
            for l, (rl, nl, hl) in enumerate(pp[5:]):
                if nl == 0:
                    continue
                if l > 2 or nl > 3:
                    raise NotImplementedError('projectors with l > 2 or more than 3 per channel')
                blocks.append((row, l, nl, np.asarray(hl, dtype=float)))
                for ii in range(nl):
                    proj_tab.append((ia, l, ii))
                    proj_rl.append(rl)
                    row += 2 * l + 1
        vlocR = be.empty((1, G))
        be.pp_local_potential(acoords, pp_par, mesh, a, vlocR)
        rcut = gto.estimate_rcut_per_shell(cell)
        Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
        ao_args = (np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut)
        coords_soa = be.to_device(np.ascontiguousarray(self.grids.coords.T))
        ur, ui = be.empty((nao, G)), be.empty((nao, G))
        out = []
        for k in kpts_lst:
            gamma = abs(k).sum() < 1e-9
            if gamma:
                be.eval_ao(*ao_args, coords_soa, ur)
                v = be.empty((1, nao, nao))
                be.vj_from_vR(ur, G, vlocR, v)
                vpp = be.to_host(v)[0].astype(np.complex128)
            else:
                be.eval_ao_k(*ao_args, k, True, coords_soa, ur, ui)
                vre, vim = be.empty((nao, nao)), be.empty((nao, nao))
                be.vj_k(ur, ui, G, vlocR, vre, vim)
                vpp = be.to_host(vre) + 1j * be.to_host(vim)
            if proj_tab:
                ov = be.empty((row, nao), dtype=torch.complex128)
                be.pp_projector_overlaps(ao_args[0], ao_args[1], ao_args[2], acoords, k, np.array(proj_tab), np.array(proj_rl),
                                         mesh, a, ov)
                S = be.to_host(ov)
                vnl = np.zeros((nao, nao), dtype=np.complex128)
                for row0, l, nl, hl in blocks:
                    deg = 2 * l + 1
                    blk = S[row0:row0 + nl * deg].reshape(nl, deg, nao)
                    vnl += np.einsum('imp,ij,jmq->pq', blk.conj(), hl, blk)
                vpp = vpp + vnl / cell.vol
            out.append(vpp.real if gamma else vpp)
        return out[0] if single else np.asarray(out)

    def get_nuc(self, kpts=None):
        """Nuclear-attraction AO matrix with the G=0 term removed, pyscf/pbc/df/fft.py:39-62:
        vne^k = ao_k^H (vneR ao_k),  vneR = ifft(coulG * sum_a (-Z_a) exp(-i G.R_a)).real.
        The potential is assembled on the host (O(G natm)); the contraction runs on the device with the
        J kernels (isdf_vj_from_vR / isdf_vj_k).  Returns (nao,nao) for a single k-point (or Gamma),
        else (nk,nao,nao), like the reference."""
        cell, be = self.cell, self.backend
        if kpts is None:
            kpts_lst, single = np.zeros((1, 3)), True
        else:
            kpts_lst = np.reshape(kpts, (-1, 3))
            single = np.ndim(kpts) == 1
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        nao = cell.nao_nr()
        Gv = cell.get_Gv(mesh)
        charge = -np.asarray(cell.atom_charges(), dtype=float)
        SI = np.exp(-1j * np.dot(cell.atom_coords(), Gv.T))
        rhoG = charge.dot(SI)
        vneG = rhoG * be.to_host(be.coulG_q(mesh, cell.lattice_vectors(), np.zeros(3)))
        vneR = np.fft.ifftn(vneG.reshape(*mesh)).real.ravel()
        d_v = be.to_device(vneR.reshape(1, G))
        rcut = gto.estimate_rcut_per_shell(cell)
        Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
        ao_args = (np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut)
        coords_soa = be.to_device(np.ascontiguousarray(self.grids.coords.T))
        out = []
        ur = be.empty((nao, G))
        ui = be.empty((nao, G))
        for k in kpts_lst:
            if abs(k).sum() < 1e-9:
                be.eval_ao(*ao_args, coords_soa, ur)
                v = be.empty((1, nao, nao))
                be.vj_from_vR(ur, G, d_v, v)
                out.append(be.to_host(v)[0])
            else:
                be.eval_ao_k(*ao_args, k, True, coords_soa, ur, ui)
                vre, vim = be.empty((nao, nao)), be.empty((nao, nao))
                be.vj_k(ur, ui, G, d_v, vre, vim)
                out.append(be.to_host(vre) + 1j * be.to_host(vim))
        return out[0] if single else np.asarray(out)
