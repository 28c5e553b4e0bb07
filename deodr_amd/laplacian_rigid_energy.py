"""``LaplacianRigidEnergy`` with the reference's interface (deodr/laplacian_rigid_energy.py:15-41), evaluated on the device."""

import numpy as np
import torch

from .scene3d import LaplacianRigidEnergyDevice


class LaplacianRigidEnergy:
    def __init__(self, mesh, vertices, cregu):
        self.mesh, self.cregu = mesh, cregu
        self.vertices_ref = np.array(vertices, dtype=np.float64)
        self._dev = LaplacianRigidEnergyDevice(mesh.adjacencies.topology, self.vertices_ref, cregu)
        self._hessian = None

    @property
    def approx_hessian(self):
        """cregu * (L^T L  kron  I3) as a SciPy CSR matrix (only built when somebody asks for it)"""
        if self._hessian is None:
            import scipy.sparse as sp

            t = self._dev.topology
            m = sp.coo_matrix((t._m_vals.cpu().numpy(), (t._m_rows.cpu().numpy(), t._m_cols.cpu().numpy())), shape=(t.nb_vertices,) * 2)
            self._hessian = (self.cregu * sp.kron(m, sp.eye(3))).tocsr()
        return self._hessian

    def evaluate(self, vertices):
        v = torch.as_tensor(np.asarray(vertices, dtype=np.float64), device=self._dev.topology.device)
        energy, grad = self._dev.evaluate(v)
        return float(energy), grad.cpu().numpy(), self.approx_hessian
