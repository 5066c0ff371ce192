"""Vectorised problem -> Ising encoders that emit dense or CSR couplings directly.

The reference builds J and h one element at a time through `IsingModel.set_coupling`, which
on its default sparse model densifies and re-sparsifies the whole matrix per call
(core/ising_model.py:94-99) -- O(n^2) per term, unusable beyond n ~ 10^3 -- and
`ConstraintEncoder._apply_constraint_to_model` (core/constraints.py:360-377) accumulates on
dense models but OVERWRITES on sparse ones (:376).  Here every constraint family is expanded
with array operations into COO triples and assembled once.

Two conventions, chosen per builder:

* ``convention="reference"`` reproduces what the reference writes, term for term: penalty
  coefficients are ADDED to `external_fields` / couplings (constraints.py:366-377), the
  linear coefficient of an equality penalty is lambda*(c_i^2 - 2 t c_i) (:75-78), and
  ``overwrite=True`` gives its sparse-model last-write-wins behaviour.  Pinned against
  tests/golden/encoders.npz.
* ``convention="physical"`` produces the Ising model whose energy
  H(s) = -1/2 sum J_ij s_i s_j - sum h_i s_i equals objective + penalties up to the tracked
  constant (`builder.constant`): signs follow H's minus signs and c_i^2 s_i^2 is a constant,
  not a field.  This is the one to anneal; the C2b / C4 / C5 style instances use it.
"""
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np


class IsingBuilder:
    def __init__(self, n_spins: int, convention: str = "physical", overwrite: bool = False):
        if convention not in ("physical", "reference"):
            raise ValueError("convention must be 'physical' or 'reference'")
        self.n = int(n_spins)
        self.convention = convention
        self.overwrite = bool(overwrite)
        self.h = np.zeros(self.n, np.float64)
        self.constant = 0.0
        self._rows, self._cols, self._vals = [], [], []

    # ------------------------------------------------------------------ raw terms
    def add_field(self, idx, coeff) -> None:
        """h[idx] += coeff (duplicates accumulate)."""
        np.add.at(self.h, np.asarray(idx, np.int64), np.asarray(coeff, np.float64))

    def add_coupling(self, i, j, coeff) -> None:
        """J[i,j] and J[j,i] += coeff (or = coeff with overwrite), arrays allowed."""
        i, j = np.asarray(i, np.int64).ravel(), np.asarray(j, np.int64).ravel()
        v = np.broadcast_to(np.asarray(coeff, np.float64), i.shape).ravel()
        if i.size and (i.min() < 0 or j.min() < 0 or i.max() >= self.n or j.max() >= self.n):
            raise ValueError("spin index out of range")
        self._rows.append(np.minimum(i, j))
        self._cols.append(np.maximum(i, j))
        self._vals.append(v.copy())

    # ------------------------------------------------------------------ constraint families
    def add_equality(self, spins, coefficients, target: float, weight: float = 1.0) -> None:
        """Penalty weight * (sum_i c_i s_i - target)^2  (reference constraints.py:51-92)."""
        s = np.asarray(spins, np.int64)
        c = np.asarray(coefficients, np.float64)
        a, b = np.triu_indices(s.size, 1)
        pair = 2.0 * weight * c[a] * c[b]
        if self.convention == "reference":
            self.add_field(s, weight * (c ** 2 - 2.0 * target * c))
            self.add_coupling(s[a], s[b], pair)
        else:
            # w (sum c s - t)^2 = w [sum c^2 + t^2] + sum_{a<b} 2w c_a c_b s_a s_b - 2 w t sum c s
            # and H = -sum_{a<b} J_ab s_a s_b - sum h s  =>  J = -pair, h = +2 w t c
            self.add_field(s, 2.0 * weight * target * c)
            self.add_coupling(s[a], s[b], -pair)
            self.constant += weight * (float(np.sum(c ** 2)) + target * target)

    def add_inequality(self, spins, coefficients, target: float, weight: float = 1.0) -> None:
        """The reference treats <= as the equality penalty (constraints.py:118-125)."""
        self.add_equality(spins, coefficients, target, weight)

    def add_cardinality(self, spins, k: int, weight: float = 1.0) -> None:
        """Exactly k of `spins` are +1: (sum (1+s)/2 - k)^2 (constraints.py:147-158)."""
        s = np.asarray(spins, np.int64)
        self.add_equality(s, np.ones(s.size), 2 * k - s.size, weight / 4.0)

    def add_cardinality_groups(self, groups, k: int, weight: float = 1.0) -> None:
        """`groups` [G, m]: the same cardinality penalty on every row, in one shot."""
        g = np.asarray(groups, np.int64)
        if g.ndim != 2:
            raise ValueError("groups must be [G, m]")
        G, m = g.shape
        w, target = weight / 4.0, float(2 * k - m)
        a, b = np.triu_indices(m, 1)
        if self.convention == "reference":
            self.add_field(g.ravel(), np.full(g.size, w * (1.0 - 2.0 * target)))
            self.add_coupling(g[:, a].ravel(), g[:, b].ravel(), 2.0 * w)
        else:
            self.add_field(g.ravel(), np.full(g.size, 2.0 * w * target))
            self.add_coupling(g[:, a].ravel(), g[:, b].ravel(), -2.0 * w)
            self.constant += G * w * (m + target * target)

    def add_qubo_pair(self, a, b, q) -> None:
        """Objective term q * x_a x_b with x = (1+s)/2 (physical convention only)."""
        if self.convention != "physical":
            raise ValueError("QUBO terms are defined for the physical convention")
        q = np.broadcast_to(np.asarray(q, np.float64), np.asarray(a).shape)
        self.add_coupling(a, b, -q / 4.0)
        self.add_field(a, -q / 4.0)
        self.add_field(b, -q / 4.0)
        self.constant += float(np.sum(q)) / 4.0

    def add_qubo_linear(self, a, q) -> None:
        """Objective term q * x_a (physical convention only)."""
        if self.convention != "physical":
            raise ValueError("QUBO terms are defined for the physical convention")
        q = np.broadcast_to(np.asarray(q, np.float64), np.asarray(a).shape)
        self.add_field(a, -q / 2.0)
        self.constant += float(np.sum(q)) / 2.0

    # ------------------------------------------------------------------ assembly
    def _triples(self):
        if not self._rows:
            z = np.zeros(0, np.int64)
            return z, z, np.zeros(0)
        r, c, v = np.concatenate(self._rows), np.concatenate(self._cols), np.concatenate(self._vals)
        key = r * self.n + c
        if self.overwrite:  # last write wins (reference sparse path, constraints.py:376)
            _, first_of_reversed = np.unique(key[::-1], return_index=True)
            keep = np.sort(key.size - 1 - first_of_reversed)
            return r[keep], c[keep], v[keep]
        uniq, inv = np.unique(key, return_inverse=True)
        acc = np.zeros(uniq.size)
        np.add.at(acc, inv, v)
        return uniq // self.n, uniq % self.n, acc

    def to_csr(self):
        """(rowptr int32 [n+1], colidx int32 [nnz], val fp32 [nnz]), both triangles, sorted."""
        import scipy.sparse as sp
        if self.overwrite:
            r, c, v = self._triples()
        else:  # duplicates are summed by the COO -> CSR conversion
            z = np.zeros(0, np.int64)
            r = np.concatenate(self._rows) if self._rows else z
            c = np.concatenate(self._cols) if self._cols else z
            v = np.concatenate(self._vals) if self._vals else np.zeros(0)
        up = sp.coo_matrix((v, (r, c)), shape=(self.n, self.n)).tocsr()
        full = up + sp.triu(up, 1).T
        full = full.tocsr()
        full.eliminate_zeros()
        full.sort_indices()
        return (full.indptr.astype(np.int32), full.indices.astype(np.int32),
                full.data.astype(np.float32))

    def to_dense(self) -> np.ndarray:
        r, c, v = self._triples()
        J = np.zeros((self.n, self.n), np.float64)
        J[r, c] = v
        J[c, r] = v
        return J.astype(np.float32)

    def fields(self) -> np.ndarray:
        return self.h.astype(np.float32)

    def to_model(self, sparse: Optional[bool] = None):
        """An IsingModel carrying these couplings (sparse COO above 2048 spins by default)."""
        import torch
        from .ising_model import IsingModel, IsingModelConfig
        sparse = (self.n > 2048) if sparse is None else sparse
        m = IsingModel(IsingModelConfig(n_spins=self.n, use_sparse=sparse))
        if sparse:
            rowptr, col, val = self.to_csr()
            rows = np.repeat(np.arange(self.n), np.diff(rowptr))
            idx = torch.from_numpy(np.stack([rows, col]).astype(np.int64))
            m.couplings = torch.sparse_coo_tensor(idx, torch.from_numpy(val), (self.n, self.n)).coalesce()
        else:
            m.couplings = torch.from_numpy(self.to_dense())
        m.set_external_fields(torch.from_numpy(self.fields()))
        return m

    def penalty_energy_offset(self) -> float:
        """H(s) + offset == objective(s) + penalties(s) in the physical convention."""
        return self.constant


# ---------------------------------------------------------------------------------------
# problem encoders
# ---------------------------------------------------------------------------------------
def tsp_ising(distance_matrix, city_visit: float = 100.0, position_fill: float = 100.0,
              convention: str = "physical", overwrite: bool = False,
              auto_scale: bool = True) -> IsingBuilder:
    """Position-based TSP encoding, spin (city, position) -> city * n + position
    (reference problems/routing.py:193-328): tour-length couplings between consecutive
    positions, one-hot penalties per city and per position."""
    d = np.asarray(distance_matrix, np.float64)
    n = d.shape[0]
    if d.shape != (n, n) or n < 2:
        raise ValueError("distance_matrix must be square with at least 2 cities")
    if auto_scale and n > 50:  # routing.py:237-241
        f = np.sqrt(n / 50.0)
        city_visit, position_fill = city_visit * f, position_fill * f
    b = IsingBuilder(n * n, convention, overwrite)
    ci, cj = np.nonzero(~np.eye(n, dtype=bool))          # ordered city pairs, row-major (:277-279)
    pos = np.arange(n)
    a_idx = (ci[:, None] * n + pos[None, :]).ravel()       # (city_i, p)
    b_idx = (cj[:, None] * n + (pos[None, :] + 1) % n).ravel()  # (city_j, p+1)
    dist = np.repeat(d[ci, cj], n)
    if convention == "reference":
        b.add_coupling(a_idx, b_idx, -dist)                # "current_coupling - distance" (:292)
    else:
        b.add_qubo_pair(a_idx, b_idx, dist)                # tour length = sum d_ij x_ip x_j,p+1
    grid = np.arange(n * n).reshape(n, n)
    b.add_cardinality_groups(grid, 1, city_visit)          # each city once (:295-311)
    b.add_cardinality_groups(grid.T, 1, position_fill)     # each position once (:313-328)
    return b


def tsp_csr(distance_matrix, city_visit: float = 100.0, position_fill: float = 100.0,
            auto_scale: bool = True, device=None):
    """The CSR couplings of `tsp_ising(..., convention="physical")` written row by row from
    the structure of the encoding instead of assembled from COO triples: every spin
    (city c, position p) has exactly 4(n-1) neighbours -- the other positions of its city, and
    for every other city the positions p-1, p, p+1 -- so rowptr is an arithmetic progression
    and each city's block of rows is three tensor writes.  1000 cities (BASELINE config 5:
    10^6 spins, 4e9 entries, 32 GB) is ~1000 small tensor programs on the device, where the
    triple route needs hundreds of GB of host memory.

    Returns torch tensors on `device` (default: CPU): rowptr int64 [n^2+1], colidx int32,
    val float32 (columns ascending within a row), h float32 [n^2], and the constant with
    H(s) + constant == tour length + penalties.  Identical to the builder's output
    (tests/test_encoders.py)."""
    import torch
    dev = torch.device("cpu") if device is None else torch.device(device)
    d = torch.as_tensor(np.asarray(distance_matrix, np.float64), device=dev)
    n = d.shape[0]
    if d.shape != (n, n) or n < 3:
        raise ValueError("distance_matrix must be square with at least 3 cities")
    if auto_scale and n > 50:  # routing.py:237-241
        f = np.sqrt(n / 50.0)
        city_visit, position_fill = city_visit * f, position_fill * f
    deg = 4 * (n - 1)
    N = n * n
    rowptr = torch.arange(N + 1, dtype=torch.int64, device=dev) * deg
    colidx = torch.empty(N * deg, dtype=torch.int32, device=dev)
    val = torch.empty(N * deg, dtype=torch.float32, device=dev)
    p = torch.arange(n, device=dev)
    # the three positions coupled to position p in another city, ascending, and what each is:
    # 0 = p-1 (that city precedes: d[c', c]), 1 = p (position one-hot), 2 = p+1 (d[c, c'])
    trip = torch.stack([(p - 1) % n, p, (p + 1) % n], 1)              # [n, 3]
    trip, kind = torch.sort(trip, 1)
    # other positions of the same city, ascending
    q = torch.arange(n - 1, device=dev)
    same_pos = q[None, :] + (q[None, :] >= p[:, None]).long()           # [n, n-1]
    same_val = torch.full((n, n - 1), -city_visit / 2.0, dtype=torch.float64, device=dev)
    cities = torch.arange(n, device=dev)
    col_view, val_view = colidx.view(n, n, deg), val.view(n, n, deg)  # [city, position, entry]
    for c in range(n):
        others = torch.cat([cities[:c], cities[c + 1:]])               # [n-1]
        oc = (others[None, :, None] * n + trip[:, None, :])             # [n, n-1, 3]
        prev_v = (-d[others, c] / 4.0)[None, :, None]                   # d[c', c]
        next_v = (-d[c, others] / 4.0)[None, :, None]                   # d[c, c']
        k = kind[:, None, :]
        ov = torch.where(k == 1, torch.full_like(prev_v, -position_fill / 2.0),
                         torch.where(k == 0, prev_v, next_v)).expand(n, n - 1, 3)
        lo = 3 * c
        col_view[c, :, :lo] = oc[:, :c].reshape(n, lo).int()
        val_view[c, :, :lo] = ov[:, :c].reshape(n, lo).float()
        col_view[c, :, lo:lo + n - 1] = (c * n + same_pos).int()
        val_view[c, :, lo:lo + n - 1] = same_val.float()
        col_view[c, :, lo + n - 1:] = oc[:, c:].reshape(n, deg - lo - (n - 1)).int()
        val_view[c, :, lo + n - 1:] = ov[:, c:].reshape(n, deg - lo - (n - 1)).float()
    off = d.clone()
    off.fill_diagonal_(0.0)
    tour_h = -(off.sum(1) + off.sum(0)) / 4.0                           # per city, every position
    card = 2.0 * (city_visit / 4.0 + position_fill / 4.0) * (2.0 - n)
    h = (tour_h[:, None] + card).expand(n, n).reshape(N).float().contiguous()
    constant = float(off.sum()) * n / 4.0 + \
        n * (city_visit / 4.0 + position_fill / 4.0) * (n + (2.0 - n) ** 2)
    return rowptr, colidx, val, h, constant


def tsp_structure(distance_matrix, city_visit: float = 100.0, position_fill: float = 100.0,
                  auto_scale: bool = True):
    """What `AnnealEngine.set_tsp` needs to run the couplings of `tsp_csr` WITHOUT storing them:
    (dist float32 [n, n], city_visit, position_fill, h float32 [n^2], constant) -- the penalty
    weights after the routing.py:237-241 scaling, the fields and the constant exactly as
    `tsp_csr` computes them (same float64 expressions, rounded to float32 once)."""
    d = np.asarray(distance_matrix, np.float64)
    n = d.shape[0]
    if d.shape != (n, n) or n < 3:
        raise ValueError("distance_matrix must be square with at least 3 cities")
    if auto_scale and n > 50:  # routing.py:237-241
        f = np.sqrt(n / 50.0)
        city_visit, position_fill = city_visit * f, position_fill * f
    off = d.copy()
    np.fill_diagonal(off, 0.0)
    tour_h = -(off.sum(1) + off.sum(0)) / 4.0
    card = 2.0 * (city_visit / 4.0 + position_fill / 4.0) * (2.0 - n)
    h = np.ascontiguousarray(np.broadcast_to((tour_h[:, None] + card), (n, n)).reshape(n * n).astype(np.float32))
    constant = float(off.sum()) * n / 4.0 + \
        n * (city_visit / 4.0 + position_fill / 4.0) * (n + (2.0 - n) ** 2)
    # the couplings tsp_csr stores are float32(-w / 2) and float32(-d / 4): scaling by a power of two
    # commutes with the rounding, so float32 weights / distances reproduce them exactly
    return d.astype(np.float32), float(np.float32(city_visit)), float(np.float32(position_fill)), h, constant


def scheduling_ising(durations: Sequence[float], n_agents: int, time_horizon: float,
                     time_discretization: int, due_dates: Optional[Sequence[float]] = None,
                     priorities: Optional[Sequence[float]] = None, objective: str = "makespan",
                     penalty_weights: Optional[Dict[str, float]] = None,
                     convention: str = "physical", overwrite: bool = False) -> IsingBuilder:
    """Multi-agent scheduling, spin (task, agent, slot) -> (task*A + agent)*S + slot
    (reference problems/scheduling.py:67-285): completion-time objective on the fields,
    one-hot assignment per task, at-most-one task per (agent, slot), precedence i<j,
    due-date penalties."""
    dur = np.asarray(durations, np.float64)
    T, A, S = dur.size, int(n_agents), int(time_discretization)
    if penalty_weights is None:  # scheduling.py:86-92
        penalty_weights = {"assignment": 100.0, "capacity": 50.0, "precedence": 75.0,
                           "time_window": 60.0}
    pr = np.ones(T) if priorities is None else np.asarray(priorities, np.float64)
    step = time_horizon / S
    idx = np.arange(T * A * S).reshape(T, A, S)
    b = IsingBuilder(T * A * S, convention, overwrite)
    completion = (np.arange(S)[None, None, :] + 1) * step + dur[:, None, None]
    if objective == "makespan":                      # :151-170
        lin = 0.1 * completion
    elif objective == "total_time":                  # :172-185
        lin = completion * pr[:, None, None]
    elif objective == "weighted_completion":         # :187-201
        lin = completion * pr[:, None, None]
    else:
        raise ValueError(f"Unknown objective: {objective}")
    lin = np.broadcast_to(lin, idx.shape)
    if convention == "reference":
        b.add_field(idx.ravel(), lin.ravel())
    else:
        b.add_qubo_linear(idx.ravel(), lin.ravel())
    b.add_cardinality_groups(idx.reshape(T, A * S), 1, penalty_weights["assignment"])  # :203-219
    # capacity (:221-245): tasks whose execution window covers `slot` on one agent
    dslots = np.ceil(dur * S / time_horizon).astype(np.int64)
    for slot in range(S):
        members = []
        for t in range(T):
            lo, hi = max(0, slot - dslots[t] + 1), min(S, slot + 1)
            starts = np.arange(lo, hi)
            starts = starts[starts + dslots[t] > slot]
            members.append((np.full(starts.size, t), starts))
        tt = np.concatenate([m[0] for m in members])
        ss = np.concatenate([m[1] for m in members])
        if tt.size > 1:
            groups = idx[tt[None, :], np.arange(A)[:, None], ss[None, :]]  # [A, members]
            b.add_cardinality_groups(groups, 1, penalty_weights["capacity"])
    if "precedence" in penalty_weights:              # :247-268: i < j, start_j <= start_i
        w = penalty_weights["precedence"]
        ti, tj = np.triu_indices(T, 1)
        si, sj = np.nonzero(np.arange(S)[None, :] <= np.arange(S)[:, None])  # sj <= si
        for i, j in zip(ti, tj):
            a_idx = idx[i][:, si]                    # [A, P]
            b_idx = idx[j][:, sj]
            ia = np.repeat(a_idx, A, axis=0).ravel()       # agent_i outer, agent_j inner
            ib = np.tile(b_idx, (A, 1)).ravel()
            if convention == "reference":
                b.add_coupling(ia, ib, w)
            else:
                b.add_qubo_pair(ia, ib, w)
    if "time_window" in penalty_weights and due_dates is not None:  # :270-285
        w = penalty_weights["time_window"]
        for t, due in enumerate(due_dates):
            if due is None or (isinstance(due, float) and np.isnan(due)):
                continue
            late = idx[t][:, int(due / step):].ravel()
            if convention == "reference":
                b.add_field(late, np.full(late.size, w))
            else:
                b.add_qubo_linear(late, np.full(late.size, w))
    return b


def assignment_ising(n_agents: int, n_tasks: int, weight: float = 100.0, costs=None,
                     convention: str = "physical") -> IsingBuilder:
    """n_agents x n_tasks assignment (BASELINE configs[1] parity instance):
    spin (agent, task) -> agent * n_tasks + task, one-hot per task and per agent."""
    b = IsingBuilder(n_agents * n_tasks, convention)
    grid = np.arange(n_agents * n_tasks).reshape(n_agents, n_tasks)
    b.add_cardinality_groups(grid.T, 1, weight)  # every task taken by exactly one agent
    b.add_cardinality_groups(grid, 1, weight)    # every agent takes exactly one task
    if costs is not None:
        c = np.asarray(costs, np.float64).reshape(n_agents * n_tasks)
        if convention == "reference":
            b.add_field(np.arange(c.size), c)
        else:
            b.add_qubo_linear(np.arange(c.size), c)
    return b


def evaluate_penalties(spins, terms: Iterable[Tuple[str, tuple]]) -> float:
    """Direct evaluation of constraint violations (reference Constraint.evaluate,
    constraints.py:66-70,112-116,139-145) for checking encodings: terms are
    ("equality" | "inequality", (spins, coefficients, target, weight)) or
    ("cardinality", (spins, k, weight))."""
    s = np.asarray(spins, np.float64)
    total = 0.0
    for kind, args in terms:
        if kind == "cardinality":
            idx, k, w = args
            total += w * (float(np.sum(s[np.asarray(idx)] == 1)) - k) ** 2
        else:
            idx, c, t, w = args
            val = float(np.dot(np.asarray(c, np.float64), s[np.asarray(idx)]))
            total += w * (max(0.0, val - t) ** 2 if kind == "inequality" else (val - t) ** 2)
    return total
