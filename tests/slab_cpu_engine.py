"""CPU stand-in for the HIP engine's ``*_planes`` entry points, built on the oracle.  Test
infrastructure: lets the z-slab driver (decomposition, ghost exchange, schedule) run with the
gloo backend on machines without a GPU.  Slab layout ``[q][nz_local + 2][ny][nx]``."""
import torch

from oracle import lettuce_oracle as orc


class OracleSlabEngine:
    def __init__(self, lattice_name, dtype, collision):
        self.lat = orc.LATTICES[lattice_name]
        self.e, self.w = orc.lattice_tensors(self.lat, dtype)
        self.collision = collision

    def _collide(self, f, tau):
        if self.collision == "bgk":
            return orc.bgk(f, tau, self.e, self.w)
        if self.collision == "kbc":
            return orc.kbc(f, tau, self.e, self.w)
        return f

    def _pull(self, f, b, e):
        """post-streaming populations of planes [b, e): f_q(x) = f*_q(x - e_q); periodic in
        x and y, ghost planes supply z."""
        out = torch.empty_like(f[:, b:e])
        for q in range(self.lat.q):
            ex, ey, ez = self.lat.e[q]
            moved = torch.roll(f[q], shifts=(ey, ex), dims=(1, 2))
            out[q] = moved[b - ez:e - ez]
        return out

    def collide_planes(self, f, out, tau, b, e):
        out[:, b:e] = self._collide(f[:, b:e], tau)

    def stream_planes(self, f, out, b, e):
        out[:, b:e] = self._pull(f, b, e)

    def stream_collide_planes(self, f, out, tau, b, e):
        out[:, b:e] = self._collide(self._pull(f, b, e), tau)
