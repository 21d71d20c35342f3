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
        self.entries, self.ncm, self.nsm = [], None, None
        self.ghosts = 1                   # ghost planes per side of the slab tensors (set by the driver's GHOST)

    def set_boundaries(self, entries, ncm, nsm, units):
        """entries as for lettuce_amd._native.Plan; masks in slab layout [nz+2, ny, nx] /
        [q, nz+2, ny, nx]."""
        self.entries, self.ncm, self.nsm = entries, ncm, nsm

    def _boundaries(self, g, ncm, lo, hi):
        """g: [q, x, y, zsub] (reference axis order), ncm: [x, y, zsub], planes [lo, hi) of the slab;
        boundaries in index order, each seeing the current field (lettuce/_simulation.py:186-188)."""
        for idx, b in enumerate(self.entries, start=1):
            if b["kind"] == "bounce_back":
                new = orc.bounce_back(g, self.lat)
            elif b["kind"] == "equilibrium" and "field" in b:
                new = b["field"][:, lo:hi].permute(0, 3, 2, 1).to(g.dtype)     # [q, z, y, x] -> [q, x, y, zsub]
            elif b["kind"] == "equilibrium":
                feq = torch.tensor(b["feq"], dtype=g.dtype).reshape(-1, 1, 1, 1)
                new = feq * torch.ones_like(g)
            elif b["axis"] == 2:
                # outlet along the decomposed axis: on the rank that holds it, the plane and its neighbour are
                # the last / first two interior planes of the slab; the block [lo, hi) was extended to hold both
                if not b.get("present", True):
                    continue
                n2 = self.ncm.shape[0]
                ghosts = self.ghosts
                plane = n2 - 1 - ghosts if b["side"] > 0 else ghosts
                nbr = plane - b["side"]
                if not (lo <= plane < hi):
                    continue
                assert lo <= nbr < hi
                pair = g[..., [nbr - lo, plane - lo]] if b["side"] > 0 else g[..., [plane - lo, nbr - lo]]
                direction = [0, 0, b["side"]]
                pair = orc.abb_outlet_inplace(pair.clone(), orc.OracleBoundary("abb_outlet", direction=direction),
                                              self.lat, self.e, self.w)
                new = g.clone()
                new[..., plane - lo] = pair[..., -1 if b["side"] > 0 else 0]
                g = new              # the outlet rewrites its whole plane, whatever the nodes' indices
                continue
            else:
                direction = [0, 0, 0]
                direction[b["axis"]] = b["side"]
                new = orc.abb_outlet_inplace(g, orc.OracleBoundary("abb_outlet", direction=direction),
                                             self.lat, self.e, self.w)
            g = torch.where(ncm == idx, new, g)
        return g

    def _z_outlet_range(self, b0, e0):
        """[lo, hi) >= [b0, e0) that also holds the plane next to a z outlet inside [b0, e0)"""
        lo, hi = b0, e0
        for b in self.entries:
            if b["kind"] == "abb_outlet" and b["axis"] == 2 and b.get("present", True):
                n2 = self.ncm.shape[0]
                plane = n2 - 1 - self.ghosts if b["side"] > 0 else self.ghosts
                if b0 <= plane < e0:
                    lo, hi = min(lo, plane - b["side"]), max(hi, plane - b["side"] + 1)
        return lo, hi

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

    def _collide_and_boundaries(self, planes, tau, b, e):
        if self.ncm is None:
            return self._collide(planes, tau)
        g = planes.permute(0, 3, 2, 1).clone()                  # [q, x, y, zsub]
        ncm = self.ncm[b:e].permute(2, 1, 0)
        g = torch.where(ncm == 0, self._collide(g, tau), g)
        return self._boundaries(g, ncm, b, e).permute(0, 3, 2, 1)

    def _stream(self, f, b, e):
        pulled = self._pull(f, b, e)
        if self.nsm is None:
            return pulled
        return torch.where(self.nsm[:, b:e] == 1, f[:, b:e], pulled)

    def collide_planes(self, f, out, tau, b, e):
        lo, hi = self._z_outlet_range(b, e) if self.ncm is not None else (b, e)
        out[:, b:e] = self._collide_and_boundaries(f[:, lo:hi], tau, lo, hi)[:, b - lo:e - lo]

    def stream_planes(self, f, out, b, e):
        out[:, b:e] = self._stream(f, b, e)

    def stream_collide_planes(self, f, out, tau, b, e):
        lo, hi = self._z_outlet_range(b, e) if self.ncm is not None else (b, e)
        out[:, b:e] = self._collide_and_boundaries(self._stream(f, lo, hi), tau, lo, hi)[:, b - lo:e - lo]

    # ---- two-step slabs (two ghost planes per side) -------------------------------------------
    def stream_collide_twice_planes(self, f, out, tau, b, e):
        tmp = f.clone()
        self.stream_collide_planes(f, tmp, tau, b - 1, e + 1)
        self.stream_collide_planes(tmp, out, tau, b, e)

    def stream_collide_twice_edges_direct(self, f, out, tau, edge, recv_lower, recv_upper, pack_lower, pack_upper):
        """lt_stream_collide_twice_edges_direct: the planes beyond the cuts come from the received messages (or, with
        None, from the ghost planes of f), the outgoing messages are written by the same call"""
        src = f
        if recv_lower is not None:
            src = f.clone()
            self.unpack_two_step(src, -1, recv_lower)
            self.unpack_two_step(src, +1, recv_upper)
        lo, hi = 2, f.shape[1] - 2
        self.stream_collide_twice_planes(src, out, tau, lo, lo + edge)
        self.stream_collide_twice_planes(src, out, tau, hi - edge, hi)
        self.pack_two_step(out, -1, pack_lower)
        self.pack_two_step(out, +1, pack_upper)

    def _sets(self, direction):
        ez = [v[2] for v in self.lat.e]
        return ([q for q in range(self.lat.q) if ez[q] == 0],
                [q for q in range(self.lat.q) if ez[q] == direction])

    def two_step_admitted(self):
        """the HIP engine's rule for boundaries (lt_plan_two_step_admitted): outlets along x only"""
        for b in self.entries:
            if b["kind"] == "abb_outlet" and b["axis"] != 0:
                return "anti-bounce-back outlet along y or z"
        return None

    def pack_two_step(self, f, side, buf):
        """[in-plane of the near plane | crossing of the near plane | crossing of the far plane | with masks: the
        populations of the near plane moving away from the cut] (include/lettuce_hip.h, lt_slab_pack_two_step)"""
        n2 = f.shape[1]
        near, far = (2, 3) if side < 0 else (n2 - 3, n2 - 4)
        in_plane, cross = self._sets(side)
        parts = [f[in_plane, near], f[cross, near], f[cross, far]]
        if self.ncm is not None:
            parts.append(f[self._sets(-side)[1], near])
        buf.copy_(torch.cat(parts))

    def unpack_two_step(self, f, side, buf):
        n2 = f.shape[1]
        near, far = (1, 0) if side < 0 else (n2 - 2, n2 - 1)
        in_plane, cross = self._sets(-side)
        a, b = len(in_plane), len(cross)
        f[in_plane, near] = buf[:a]
        f[cross, near] = buf[a:a + b]
        f[cross, far] = buf[a + b:a + 2 * b]
        if self.ncm is not None:
            f[self._sets(side)[1], near] = buf[a + 2 * b:]
