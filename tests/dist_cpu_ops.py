"""CPU test double for pytorch_sparse_solver.distributed (TEST CODE, never imported by the product).

Implements the `ops` interface of distributed.py with numpy + the oracle's primitives, i.e. the
same arithmetic spec as the HIP kernels, so the multi-rank orchestration (row partition on chunk
boundaries, halo plan, all_to_all halo exchange, all_gather of chunk partials) can be verified
with the gloo backend on CPU -- including the claim that the result is bitwise identical to the
single-rank solve for any rank count."""
import numpy as np
import torch

from oracle import oracle as O

INT64_MAX = np.iinfo(np.int64).max


class OracleOps:
    def __init__(self):
        self.device = torch.device("cpu")

    def empty(self, n, dtype=None):
        return torch.zeros(n, dtype=dtype or torch.float64)

    zeros = empty

    def make_matrix(self, crow, col_local, val, n_local, n_cols_ext, ch):
        crow = crow.numpy().astype(np.int64)
        return {"crow": (crow - crow[0]).astype(np.int32), "col": col_local.numpy().astype(np.int32),
                "val": val.numpy().astype(np.float64), "n": n_local, "ch": ch}

    def free_matrix(self, m):
        pass

    def spmv(self, m, x_ext, y, mode=0, w=None, bsub=None, part0=None, part1=None, stop=None, it=0):
        if stop is not None and it >= int(stop.item()):
            return
        n, ch = m["n"], m["ch"]
        out = O.spmv(m["crow"], m["col"], m["val"], x_ext.numpy(), bsub=None if bsub is None else bsub.numpy()[:n])
        y[:n] = torch.from_numpy(out)
        if mode & 1:
            p = O.dot_tiled_parts_ch(w.numpy()[:n], out, ch)
            part0[:p.size] = torch.from_numpy(p)
        if mode & 2:
            p = O.dot_tiled_parts_ch(out, out, ch)
            part1[:p.size] = torch.from_numpy(p)

    def dot_parts(self, n, ch, x, y, part):
        p = O.dot_parts_ch(x.numpy()[:n], y.numpy()[:n], ch)
        part[:p.size] = torch.from_numpy(p)

    def reduce_parts(self, part, g):
        return torch.tensor([O.reduce_parts(part.numpy()[:g])], dtype=torch.float64)

    def gather(self, idx, src, dst):
        dst[:idx.numel()] = src[idx.long()]

    def scal_alloc(self):
        return torch.zeros(8, dtype=torch.float64)

    def stop_word(self, scal):
        return scal[6:7].view(torch.int64)

    def cg_start(self, n, ch, g, scal, part_rr, part_bb, r, p, tol, atol, maxiter):
        gamma0 = O.reduce_parts(part_rr.numpy()[:g])
        bs = O.reduce_parts(part_bb.numpy()[:g])
        tolf, atolf = np.float32(tol), np.float32(atol)
        atol2 = max(float(tolf * tolf) * bs, float(atolf * atolf))
        p[:n] = r[:n]
        scal[0], scal[1], scal[2], scal[3] = gamma0, 0.0, atol2, bs
        self.stop_word(scal)[0] = 0 if (maxiter <= 0 or gamma0 <= atol2) else INT64_MAX

    def cg_update(self, n, ch, g, scal, it, part_pAp, Ap, r, part_out):
        if it >= int(self.stop_word(scal).item()):
            return
        pAp = O.reduce_parts(part_pAp.numpy()[:g])
        alpha = float(scal[it & 1].item()) / pAp
        rn, an = r.numpy(), Ap.numpy()
        rn[:n] = rn[:n] - alpha * an[:n]
        q = O.dot_parts_ch(rn[:n], rn[:n], ch)
        part_out[:q.size] = torch.from_numpy(q)

    def cg_direction(self, n, ch, g, scal, it, maxiter, part_pAp, part_rr, r, p, x):
        if it >= int(self.stop_word(scal).item()):
            return
        pAp = O.reduce_parts(part_pAp.numpy()[:g])
        rr = O.reduce_parts(part_rr.numpy()[:g])
        gamma = float(scal[it & 1].item())
        alpha, beta = gamma / pAp, rr / gamma
        pn, rn, xn = p.numpy(), r.numpy(), x.numpy()
        xn[:n] = xn[:n] + alpha * pn[:n]
        pn[:n] = rn[:n] + beta * pn[:n]
        scal[(it + 1) & 1] = rr
        if it + 1 >= maxiter or rr <= float(scal[2].item()):
            self.stop_word(scal)[0] = it + 1

    def read_scal(self, scal):
        return {"gamma": (scal[0].item(), scal[1].item()), "atol2": scal[2].item(), "bs": scal[3].item(),
                "stop_it": int(self.stop_word(scal).item())}
