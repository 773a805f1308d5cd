"""Curvature-matrix linear operators of an empirical risk: Hessian, GGN (exact / Monte-Carlo)
and empirical Fisher, behind the reference's constructor and ``@`` contract.

Two execution paths, chosen per operator at construction:

* **native** -- fully-connected nets in fp32 on the GPU (``mlp_native.detect_mlp``): the
  GGN / EF product of every mini-batch is one call into the HIP library
  (``clo_mlp_ggn_matvec``); no autograd, no vmap, weights streamed once per pass.
* **autograd** -- any other model: the per-batch product is written with ``torch.func``
  (forward-over-reverse for the Hessian, ``jvp -> loss-Hessian -> vjp`` for GGN-type
  matrices) and ``vmap``ped over the trailing column axis, on whatever device the parameters
  live; this is the host-framework part the north star leaves in PyTorch.

Reference: ``curvlinops/_torch_base.py:817-1007`` (base class, batch loop),
``hessian.py:13-145``, ``ggn.py:17-366``, ``gradient_moments.py:15-151``.
"""

from __future__ import annotations

from collections.abc import Callable, Iterable, MutableMapping
import torch
from torch import Tensor
from torch.func import grad, jacrev, jvp, vjp, vmap
from torch.nn import BCEWithLogitsLoss, CrossEntropyLoss, Module, MSELoss

from curvlinops_amd import _hip
from curvlinops_amd.enums import FisherType
from curvlinops_amd.linop import PyTorchLinearOperator, fresh_result_buffer
from curvlinops_amd.loss_sampling import make_grad_output_fn
from curvlinops_amd.mlp_native import NativeMLP, detect_mlp, loss_kind_and_scale
from curvlinops_amd.risk import EmpiricalRiskMixin
from curvlinops_amd.utils import make_functional_loss


class CurvatureLinearOperator(EmpiricalRiskMixin, PyTorchLinearOperator):
    """Base class: ``A @ M = sum_batches norm_b * A_b @ M`` with ``norm_b = 1`` (sum) or
    ``B_b / N_data`` (mean).  Subclasses provide ``_matvec_batch`` (autograd path) and may
    enable the native path through ``_NATIVE_KIND``."""

    FIXED_DATA_ORDER: bool = False
    _NATIVE_KIND: str | None = None  # "ggn" | "ef" | None

    def __init__(
        self,
        model_func: Module | Callable[[dict[str, Tensor], Tensor | MutableMapping], Tensor],
        loss_func: Callable[[Tensor, Tensor], Tensor] | None,
        params: dict[str, Tensor],
        data: Iterable[tuple[Tensor | MutableMapping, Tensor]],
        progressbar: bool = False,
        check_deterministic: bool = True,
        num_data: int | None = None,
        num_per_example_loss_terms: int | None = None,
        batch_size_fn: Callable[[MutableMapping | Tensor], int] | None = None,
    ):
        EmpiricalRiskMixin.__init__(
            self, model_func, loss_func, params, data, progressbar=progressbar,
            batch_size_fn=batch_size_fn, num_data=num_data,
            num_per_example_loss_terms=num_per_example_loss_terms,
            check_deterministic=check_deterministic,
        )
        PyTorchLinearOperator.__init__(self, self._get_in_shape(), self._get_out_shape())
        self._native: NativeMLP | None = None
        self._native_aux: dict[int, Tensor] = {}
        self._native_flags = _hip.MLP_DEFAULT
        self.native_column_products = 0   # products that ran on the K-column kernels (clo_mlp_*_matmat); for tests
        self._init_mp()
        self._init_native()
        if check_deterministic:
            self._check_deterministic_matvec()

    # ------------------------------------------------------------------ shapes
    def _get_in_shape(self) -> list[tuple[int, ...]]:
        return [tuple(p.shape) for p in self._params.values()]

    def _get_out_shape(self) -> list[tuple[int, ...]]:
        return [tuple(p.shape) for p in self._params.values()]

    # ------------------------------------------------------------------ autograd path
    def _init_mp(self) -> None:
        """``self._mp(X, y, M_tuple)``: ``_matvec_batch`` vmapped over the trailing axis."""
        keys = list(self._params.keys())

        def one_column(X, y, v: tuple[Tensor, ...]) -> tuple[Tensor, ...]:
            out = self._matvec_batch(X, y, dict(zip(keys, v)))
            return tuple(out[k] for k in keys)

        self._mp = vmap(one_column, in_dims=(None, None, -1), out_dims=-1, randomness="same")
        self._one_column = one_column

    def _matmat_batch(self, X, y: Tensor, M: list[Tensor]) -> list[Tensor]:
        if self.SINGLE_COLUMN_DIRECT and M and M[0].shape[-1] == 1:
            # one column: the product itself, without the vmap over a trailing axis of length one (same functions, same
            # result; the batching rules of every op in three passes through the network are host time)
            return [o.unsqueeze(-1) for o in self._one_column(X, y, tuple(m[..., 0] for m in M))]
        return list(self._mp(X, y, tuple(M)))

    SINGLE_COLUMN_DIRECT = True

    def _matvec_batch(self, X, y: Tensor, v: dict[str, Tensor]) -> dict[str, Tensor]:
        raise NotImplementedError

    # ------------------------------------------------------------------ native path
    def _init_native(self) -> None:
        if self._NATIVE_KIND is None or self._loss_func is None:
            return
        structure = detect_mlp(self._model_module, self._params)
        if structure is None or loss_kind_and_scale(self._loss_func, 1, 1) is None:
            return
        _hip.load()  # a GPU fp32 MLP must run natively: fail loudly if the library is absent
        native = NativeMLP(structure, self._params)
        if self._NATIVE_KIND == "hessian" and not native.plan.hessian_supported():
            return
        native.plan.flags = self._native_flags
        self._native = native

    @property
    def native_flags(self) -> int:
        """``CLO_MLP_*`` kernel choice of this operator's single-vector products (an argument of the C
        call, see ``include/curvlinops_amd.h``): ``_hip.MLP_NO_PERSISTENT`` keeps the launch chain, e.g.
        while a collective or another long-running kernel of the caller shares the GPU."""
        return self._native_flags

    @native_flags.setter
    def native_flags(self, flags: int) -> None:
        self._native_flags = int(flags)
        if self._native is not None:
            self._native.plan.flags = self._native_flags

    @property
    def uses_native_kernels(self) -> bool:
        return self._native is not None

    # The reference recomputes everything per product and holds params / data by reference
    # (`_torch_base.py:923-944`, `gradient_moments.py:48-87`, `hessian.py:66`).  So does the native path: on every
    # product the kernels' pointer tables are checked against the live storage of `params` (`NativeMLP.is_current`),
    # the per-batch output gradients of the EF / Hessian kernels are recomputed (one forward pass of the mini-batch)
    # and copies of the data (merged or re-laid-out mini-batches) are re-made.  Nothing derived from parameter or data
    # VALUES is kept between products: autograd version counters do not see `p.data.add_()`, `p.data.copy_()` or
    # `vector_to_parameters`, so they cannot validate such a cache.  A caller that knows better opts in:
    # `op.assume_frozen = True` keeps the derived quantities until `op.refresh()` / `assume_frozen = False`.
    @property
    def assume_frozen(self) -> bool:
        """Opt-in promise that neither the parameters nor the data change between products: the native path then keeps
        the per-batch output gradients (EF, Hessian) and its merged mini-batch copies.  Default False = the reference's
        semantics (everything re-read per product).  :meth:`refresh` drops what has been kept."""
        return getattr(self, "_assume_frozen", False)

    @assume_frozen.setter
    def assume_frozen(self, value: bool) -> None:
        self._assume_frozen = bool(value)
        self.refresh()

    def refresh(self) -> None:
        """Forget every quantity derived from parameter / data values (only kept under ``assume_frozen``)."""
        self._native_aux.clear()
        self._native_flat = None

    def _aux_lookup(self, idx: int, X: Tensor, y: Tensor) -> Tensor | None:
        if not self.assume_frozen:
            return None
        hit = self._native_aux.get(idx)
        if hit is None:
            return None
        Xr, yr, value = hit
        return value if Xr is X and yr is y else None

    def _aux_store(self, idx: int, X: Tensor, y: Tensor, value: Tensor) -> Tensor:
        if self.assume_frozen:
            self._native_aux[idx] = (X, y, value)
        return value

    def _native_rebind_if_replaced(self) -> None:
        """Storage of ``params`` replaced (entries of the dict swapped for other tensors, ``p.data = t``,
        ``vector_to_parameters``, ``module.to(...)``): bind the kernels to what is there now, or leave the
        native path if the new tensors do not qualify (dtype, device, layout)."""
        nat = self._native
        if nat is not None:
            if nat.is_current(self._params):
                return
            self._native = None
            self.refresh()
        elif getattr(self, "_native_lost", None) is None:
            return   # this operator never ran natively: nothing to follow
        sig = tuple((id(p), p.data_ptr(), p.dtype) for p in self._params.values())
        if nat is None and sig == self._native_lost:
            return   # still the tensors that did not qualify
        self._init_native()
        self._native_lost = None if self._native is not None else sig

    def _native_batch_args(self, idx: int, X: Tensor, y: Tensor, X_user: Tensor | None = None):
        """(loss_kind, scale, aux) of batch ``idx`` for the native kernel; ``X`` is the prepared
        (flattened, contiguous) input, ``X_user`` the tensor object the data iterable yielded."""
        N, C = X.shape[0], self._native.s.dims[-1]
        X_user = X if X_user is None else X_user
        if self._NATIVE_KIND == "ggn":
            kind, scale = loss_kind_and_scale(self._loss_func, N, C)
            return kind, scale, None
        if self._NATIVE_KIND == "hessian":
            # exact Hessian: besides the loss Hessian (as for the GGN) the R-operator backward needs
            # the gradient of the reduced mini-batch loss w.r.t. the prediction, G [N, C]
            kind, scale = loss_kind_and_scale(self._loss_func, N, C)
            G = self._aux_lookup(idx, X_user, y)
            if G is None:
                tgt = self._native_targets(y, N, C)
                if tgt is not None:   # on the device, from the live parameters (clo_mlp_loss_grad): no host forward pass
                    red = 1.0 if self._loss_func.reduction == "sum" else (
                        1.0 / N if isinstance(self._loss_func, CrossEntropyLoss) else 1.0 / (N * C))
                    nat = self._native
                    G = nat.plan.loss_grad(nat.W, nat.b, X, tgt, kind, red)
                else:
                    with torch.enable_grad():
                        f = self._model_func(self._params, X).detach().requires_grad_(True)
                        (G,) = torch.autograd.grad(self._loss_func(f, y), f)
                G = self._aux_store(idx, X_user, y, G.contiguous())
            return kind, scale, G
        if self._NATIVE_KIND == "mc":
            # MC-GGN: H_n = (1/c) sum_m g'_nm g'_nm^T with would-be gradients drawn from the model's
            # likelihood (ggn.py:100-168).  Drawn HERE, once per mini-batch and product, from the
            # global RNG that GGNLinearOperator._matmat seeds -- the same draws as the autograd path.
            with torch.no_grad():
                g = self._mc_sampler(self._model_func(self._params, X), y)  # [N, M, C]
            c = {"mean": float(N), "sum": 1.0}[self._loss_func.reduction]
            return _hip.LOSS_RANK1, 1.0 / c, g.reshape(N, g.shape[1], C).contiguous()
        # empirical Fisher: H_n = (1/c) g_n g_n^T with g_n the UNREDUCED per-sample gradient of the loss w.r.t. the
        # prediction (gradient_moments.py:48-87).  The kernels form g_n themselves, from the prediction they hold anyway
        # and the TARGETS (`CLO_LOSS_EF_*`, csrc/mlp_loss.h): re-derived on every product like the reference does,
        # without a forward pass on the host and without anything to keep between products.
        red = self._loss_func.reduction
        c = 1.0 if red == "sum" else float(N if isinstance(self._loss_func, CrossEntropyLoss) else N * C)
        tgt = self._native_targets(y, N, C)
        if tgt is None:
            return self._ef_host_gradients(idx, X, y, X_user, N, C, c)   # (class probabilities as targets: host route)
        # (isinstance, like the native gate `loss_kind_and_scale`: a user subclass of MSELoss / CrossEntropyLoss is that loss)
        if isinstance(self._loss_func, MSELoss):
            kind = _hip.LOSS_EF_MSE
        elif isinstance(self._loss_func, CrossEntropyLoss):
            kind = _hip.LOSS_EF_CE
        elif isinstance(self._loss_func, BCEWithLogitsLoss):
            kind = _hip.LOSS_EF_BCE
        else:
            return self._ef_host_gradients(idx, X, y, X_user, N, C, c)
        return kind, 1.0 / c, tgt.reshape(N, 1, -1)

    def _native_targets(self, y: Tensor, N: int, C: int) -> Tensor | None:
        """The targets as the kernels read them (csrc/mlp_loss.h): ``[N, C]`` fp32 for MSE / BCE -- the caller's tensor
        itself when it already is --, class labels as ``[N]`` floats for CE; None if ``y`` is something else."""
        if isinstance(self._loss_func, CrossEntropyLoss):
            if y.dim() != 1 or y.dtype.is_floating_point or y.shape[0] != N:
                return None
            # labels equal to `ignore_index` (default -100) contribute NO gradient in torch; the kernels know no such
            # rows (g = softmax - onehot): such batches take the host route.  Checked once per label tensor version.
            key = (y.data_ptr(), tuple(y.shape), y._version)
            seen = self.__dict__.setdefault("_ignored_label_checks", {})
            if key not in seen:
                if len(seen) > 64:
                    seen.clear()
                seen[key] = bool((y == self._loss_func.ignore_index).any())
            if seen[key]:
                return None
            return y.to(torch.float32)
        if y.numel() != N * C:
            return None
        t = y.reshape(N, C)
        return t if t.dtype == torch.float32 and t.is_contiguous() else t.to(torch.float32).contiguous()

    def _ef_host_gradients(self, idx, X, y, X_user, N, C, c):
        """Fallback of the EF product: per-sample output gradients from a forward pass on the host (`CLO_LOSS_RANK1`)."""
        aux = self._aux_lookup(idx, X_user, y)
        if aux is None:
            with torch.enable_grad():
                f = self._model_func(self._params, X).detach().requires_grad_(True)
                lf = type(self._loss_func)(reduction="sum")
                (g,) = torch.autograd.grad(lf(f, y), f)
            aux = self._aux_store(idx, X_user, y, g.contiguous().unsqueeze(1))  # [N, 1, C]
        return _hip.LOSS_RANK1, 1.0 / c, aux

    _MERGE_MAX_ROWS = 1024

    @staticmethod
    def _native_cost(n: int) -> float:
        """Relative time of one native product over ``n`` rows, re-fitted at the end of round 6 on C2
        (profiles/r06_c2_batch_sweep.txt: 8 rows 41 us on the persistent kernel; 9-16 rows 55-56 us, 17-32 rows 68-74 us,
        33-48 rows 88-94 us and 49-64 rows 105-107 us on the MFMA chain; beyond that the GEMM path, 151 us at 65 rows,
        172 at 128, 300 at 256, 500 at 512, 940 at 1024), in units of the 8-row product."""
        if n <= 8:
            return 1.0
        if n <= 16:
            return 1.36
        if n <= 32:
            return 1.78
        if n <= 48:
            return 2.25
        if n <= 64:
            return 2.60
        if n <= 128:
            return 3.1 + n / 120.0
        return 2.1 + n / 49.5

    def _merge_native_batches(self, entries: list[tuple]) -> list[tuple]:
        """``entries``: ``(X, kind, scale, aux, norm)`` per mini-batch.  The curvature is a sum over
        data in which every row carries the same weight ``scale * norm`` (the batch mean times
        ``B / N_data``, or the plain sum), so consecutive mini-batches can be processed as ONE larger
        batch -- same result up to the order of the floating-point sums, a fraction of the launches
        (two batches of 64 rows: 2 x 150 us vs 192 us for 128 rows).  Groups are formed greedily up to
        ``_MERGE_MAX_ROWS`` rows and kept only where the cost model says they pay off."""
        if len(entries) < 2:
            return entries
        out: list[tuple] = []
        group: list[tuple] = []

        def flush() -> None:
            if len(group) >= 2 and self._native_cost(sum(e[0].shape[0] for e in group)) < sum(
                    self._native_cost(e[0].shape[0]) for e in group):
                X = torch.cat([e[0] for e in group])
                kind, weight = group[0][1], group[0][2] * group[0][4]
                if group[0][3] is None:
                    aux = None
                elif self._NATIVE_KIND == "hessian":  # gradients of the REDUCED batch losses: weight them here
                    aux = torch.cat([e[3] * e[4] for e in group])
                else:
                    aux = torch.cat([e[3] for e in group])
                out.append((X, kind, weight, aux, 1.0))
            else:
                out.extend(group)
            group.clear()

        for e in entries:
            X, kind, scale, aux, norm = e
            compatible = bool(group) and (
                kind == group[0][1]
                and abs(scale * norm - group[0][2] * group[0][4]) <= 1e-6 * abs(group[0][2] * group[0][4])
                and (aux is None) == (group[0][3] is None)
                and (aux is None or aux.shape[1:] == group[0][3].shape[1:])
                and sum(g[0].shape[0] for g in group) + X.shape[0] <= self._MERGE_MAX_ROWS
            )
            if group and not compatible:
                flush()
            group.append(e)
        flush()
        return out

    def _matmat_native(self, M: list[Tensor]) -> list[Tensor] | None:
        """All columns of ``M`` through the HIP kernels; None if some batch does not qualify
        (then nothing has been written and the caller uses the autograd path)."""
        nat = self._native
        if nat is None:
            return None
        batches, origs = [], []
        for X, y in self._loop_over_data(desc="_matmat"):
            Xn = nat.prepare_input(X)
            if Xn is None or y.shape[0] != Xn.shape[0]:
                return None
            if Xn.shape[0] == 0:
                continue  # an empty mini-batch contributes nothing to the sum over data
            batches.append((Xn, y, self._get_normalization_factor(X, y)))
            origs.append(X)
        K = M[0].shape[-1]
        # per-batch curvature arguments ONCE per product and in data order (MC draws its samples here)
        bargs = [self._native_batch_args(bi, Xn, y, origs[bi]) for bi, (Xn, y, _) in enumerate(batches)]
        out = self._matmat_native_cols(M, batches, bargs, K)
        if out is not None:
            return out
        # K-major contiguous copies so that every column is a parameter-shaped contiguous view
        Vk = [m.movedim(-1, 0).contiguous().float() for m in M]
        Ok = [torch.empty_like(v) for v in Vk]
        if not batches:
            for o in Ok:
                o.zero_()
        merged = self._merge_native_batches([(Xn, *bargs[bi], norm) for bi, (Xn, _, norm) in enumerate(batches)])
        for k in range(K):
            V = [v[k] for v in Vk]
            O = [o[k] for o in Ok]
            for bi, (Xn, kind, scale, aux, norm) in enumerate(merged):
                if self._NATIVE_KIND == "hessian":
                    nat.hessian_matvec(V, O, Xn, aux, kind, scale, alpha=norm, beta=0.0 if bi == 0 else 1.0)
                else:
                    nat.matvec(V, O, Xn, kind, scale, alpha=norm, beta=0.0 if bi == 0 else 1.0, aux=aux)
        return [o.movedim(0, -1) for o in Ok]

    _NATIVE_COLS_MAX_ROWS = 32  # the K-column kernels run 8-row passes; beyond that GEMMs win
    _NATIVE_COLS_MIN_K = 8      # below, K matvecs on K-major copies are as fast (measured on C2)

    def _matmat_native_cols(self, M: list[Tensor], batches, bargs, K: int) -> list[Tensor] | None:
        """K >= 4 columns in the reference's K-trailing layout through ``clo_mlp_ggn_matmat`` (the
        tangent weights and the result are streamed once per column, W is shared); None if the
        shapes / layout do not qualify."""
        nat = self._native
        plan = nat.plan
        if K < self._NATIVE_COLS_MIN_K or not batches or any(b[0].shape[0] > self._NATIVE_COLS_MAX_ROWS for b in batches):
            return None
        if self._NATIVE_KIND == "hessian":
            # exact Hessian columns (clo_mlp_hessian_matmat): one contiguous block of <= 64 columns per call
            if (K % 4 or not all(m.is_contiguous() and m.data_ptr() % 16 == 0 for m in M)
                    or not plan.hessian_matmat_supported(min(K, plan.MATMAT_MAX_K), min(K, plan.MATMAT_MAX_K))
                    or any(a[0] not in (0, 1, 2) or a[2] is None for a in bargs)):
                return None
            out = self._alloc_cols_like(M)
            self.native_column_products += 1
            with torch.cuda.device(self.device):
                return self._hessian_native_cols_run(M, out, batches, bargs, K)
        # rows of the [D, K] matrix must be float4-complete: K % 4 == 0 (else the column loop runs)
        rank = max([1] + [a[2].shape[1] for a in bargs if a[2] is not None])
        if not all(m.is_contiguous() and m.data_ptr() % 16 == 0 for m in M) or not plan.matmat_supported(4, K, rank):
            return None
        out = self._alloc_cols_like(M)
        self.native_column_products += 1
        with torch.cuda.device(self.device):  # kernels launch on the operands' device, whatever is current
            return self._matmat_native_cols_run(M, out, batches, bargs, K)

    def _matmat_native_cols_run(self, M, out, batches, bargs, K: int) -> list[Tensor]:
        nat = self._native
        plan = nat.plan
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for k0 in range(0, K, plan.MATMAT_MAX_K):
            kc = min(plan.MATMAT_MAX_K, K - k0)
            ws = plan.matmat_workspace(kc, self.device)
            ptr = lambda lst, i: None if i is None else lst[i].data_ptr() + 4 * k0  # noqa: E731
            vw = [ptr(M, i) for i in nat.w_idx]
            vb = [ptr(M, i) for i in nat.b_idx]
            ow = [ptr(out, i) for i in nat.w_idx]
            ob = [ptr(out, i) for i in nat.b_idx]
            for bi, (Xn, y, norm) in enumerate(batches):
                kind, scale, aux = bargs[bi]
                plan.ggn_matmat_ptrs(vw, vb, ow, ob, K, kc, Xn.data_ptr(), Xn.shape[0], kind, scale, norm,
                                     0.0 if bi == 0 else 1.0, None if aux is None else aux.data_ptr(),
                                     1 if aux is None else aux.shape[1], ws.data_ptr(), stream)
        return out

    def _hessian_native_cols_run(self, M, out, batches, bargs, K: int) -> list[Tensor]:
        nat = self._native
        plan = nat.plan
        stream = torch.cuda.current_stream(self.device).cuda_stream
        whole = K <= plan.MATMAT_MAX_K
        for k0 in range(0, K, plan.MATMAT_MAX_K):
            kc = min(plan.MATMAT_MAX_K, K - k0)
            # the kernel wants its K columns contiguous: blocks of a wider matrix travel as copies
            Mc = M if whole else [m[..., k0:k0 + kc].contiguous() for m in M]
            Oc = out if whole else [torch.empty_like(m) for m in Mc]
            ws = plan.hessian_matmat_workspace(kc, self.device)
            ptr = lambda lst, i: None if i is None else lst[i].data_ptr()  # noqa: E731
            vw, vb = [ptr(Mc, i) for i in nat.w_idx], [ptr(Mc, i) for i in nat.b_idx]
            ow, ob = [ptr(Oc, i) for i in nat.w_idx], [ptr(Oc, i) for i in nat.b_idx]
            for bi, (Xn, y, norm) in enumerate(batches):
                kind, scale, G = bargs[bi]
                G = G.contiguous()
                plan.hessian_matmat_ptrs(vw, vb, ow, ob, kc, Xn.data_ptr(), Xn.shape[0], G.data_ptr(), kind, scale,
                                         norm, 0.0 if bi == 0 else 1.0, ws.data_ptr(), stream)
            if not whole:
                for o, oc in zip(out, Oc):
                    o[..., k0:k0 + kc] = oc
        return out

    @staticmethod
    def _alloc_cols_like(M: list[Tensor]) -> list[Tensor]:
        """Result blocks as views of ONE ``[D, K]`` buffer in ``params`` order, so that the flat
        result the base class builds with ``cat`` is already laid out."""
        K = M[0].shape[-1]
        rows = [m.numel() // K for m in M]
        buf = fresh_result_buffer(sum(rows), K, M[0].device, M[0].dtype)
        out, pos = [], 0
        for m, r in zip(M, rows):
            out.append(buf[pos:pos + r].view(m.shape))
            pos += r
        return out

    # ------------------------------------------------------------------ flat fast path
    def _native_flat_key(self) -> tuple | None:
        """What the kept flat-path ARGUMENT TABLES (addresses, row counts, loss constants) were derived from: the batch
        tensor objects, their storage addresses and shapes.  Values are not part of it -- nothing value-derived is kept
        unless ``assume_frozen``."""
        if not isinstance(self._data, (list, tuple)):
            return None
        try:
            return tuple((X, X.data_ptr(), X.shape, y, y.data_ptr(), y.shape) for X, y in self._data)
        except (AttributeError, TypeError, ValueError):
            return None

    @staticmethod
    def _same_flat_key(a: tuple | None, b: tuple | None) -> bool:
        if a is None or b is None or len(a) != len(b):
            return False
        return all(p[0] is q[0] and p[3] is q[3] and p[1] == q[1] and p[4] == q[4] and p[2] == q[2] and p[5] == q[5]
                   for p, q in zip(a, b))

    def _native_flat_setup(self):
        """Per-batch argument tables of the native path when ``data`` is a list of device-resident fp32 batches; None
        otherwise.  They are kept between products only while they hold nothing but ADDRESSES of the caller's live
        tensors (the GGN on contiguous batches: the kernels then read parameters and data in place, like the
        reference); tables that contain value-derived tensors -- the EF's per-sample output gradients, merged or
        re-laid-out copies of mini-batches -- are rebuilt on every product unless ``assume_frozen``."""
        self._native_rebind_if_replaced()
        if (self._native is None or self._NATIVE_KIND not in ("ggn", "ef")
                or not isinstance(self._data, (list, tuple))):
            return None  # the flat path is the whole-network GGN-type kernel only
        key = self._native_flat_key()
        cached = getattr(self, "_native_flat", None)
        if cached is not None and (cached[2] or self.assume_frozen) and self._same_flat_key(cached[0], key):
            return cached[1]
        self._native_flat = None
        if key is None:
            return None
        entries, live = [], True
        for bi, (X, y) in enumerate(self._data):
            if not (isinstance(X, Tensor) and X.device == self.device and y.device == self.device):
                return None
            Xn = self._native.prepare_input(X)
            if Xn is None or y.shape[0] != Xn.shape[0] or Xn.shape[0] == 0:
                return None
            live = live and Xn.data_ptr() == X.data_ptr()
            entries.append((Xn, *self._native_batch_args(bi, Xn, y, X), self._get_normalization_factor(X, y)))
        if not entries:
            return None
        # consecutive mini-batches as one larger batch where that pays off (concatenated copies)
        merged = self._merge_native_batches(entries)
        ys = {y.data_ptr() for _, y in self._data}
        live = live and len(merged) == len(entries) and all(e[3] is None or e[3].data_ptr() in ys for e in merged)
        batches = [(Xn, Xn.data_ptr(), Xn.shape[0], kind, scale, None if aux is None else aux.data_ptr(),
                    1 if aux is None else aux.shape[1], aux, norm)
                   for Xn, kind, scale, aux, norm in merged]
        nmax = max(b[2] for b in batches)
        ws = self._native.plan.workspace(nmax, self.device)
        self._native_flat = (key, (batches, ws, ws.data_ptr()), live)
        return self._native_flat[1]

    def __matmul__(self, X):
        """Fast path for a flat fp32 GPU vector on the native kernels: one output allocation and
        one C call per mini-batch, no per-parameter tensor views.  Everything else takes the
        generic route of :class:`PyTorchLinearOperator`."""
        if (
            self._native is not None
            and isinstance(X, Tensor)
            and X.dim() == 1
            and X.is_cuda
            and X.dtype == torch.float32
            and X.shape[0] == self._native.D
            and X.is_contiguous()
            and getattr(self, "_mc_samples", 0) == 0
        ):
            setup = self._native_flat_setup()
            if setup is not None:
                batches, _ws, ws_ptr = setup
                nat = self._native
                out = torch.empty_like(X)
                vp, op = X.data_ptr(), out.data_ptr()
                dev = X.device
                foreign = dev.index != torch.cuda.current_device()
                if foreign:  # operator on another GPU than the current one: launch with ITS device current
                    prev = torch.cuda.current_device()
                    torch.cuda.set_device(dev)
                try:
                    stream = torch.cuda.current_stream(dev).cuda_stream
                    beta = 0.0
                    for (_Xn, xptr, N, kind, scale, auxp, auxr, _aux, norm) in batches:
                        nat.plan.ggn_matvec_flat(vp, op, nat.w_off, nat.b_off, xptr, N, kind, scale, norm, beta,
                                                 auxp, auxr, ws_ptr, stream)
                        beta = 1.0
                finally:
                    if foreign:
                        torch.cuda.set_device(prev)
                return out
        return super().__matmul__(X)

    # ------------------------------------------------------------------ product
    def _matmat(self, M: list[Tensor]) -> list[Tensor]:
        self._native_rebind_if_replaced()
        if self._native is not None and all(m.is_cuda and m.dtype == torch.float32 for m in M):
            out = self._matmat_native(M)
            if out is not None:
                return out
        AM = [torch.zeros_like(m) for m in M]
        for X, y in self._loop_over_data(desc="_matmat"):
            norm = self._get_normalization_factor(X, y)
            for acc, cur in zip(AM, self._matmat_batch(X, y, M)):
                acc.add_(cur, alpha=norm)
        return AM


# ------------------------------------------------------------------------------------------
# per-batch products written with torch.func
# ------------------------------------------------------------------------------------------
def make_ggn_vector_product(f: Callable, c: Callable) -> Callable:
    """``(params, X, loss_args, v) -> J^T (nabla_f^2 c) J v`` for model ``f(params, X)`` and
    criterion ``c(prediction, loss_args)`` (reference ``ggn.py:17-74``)."""

    @torch.no_grad()
    def ggn_vp(params: dict[str, Tensor], X, loss_args: tuple, v: dict[str, Tensor]) -> dict[str, Tensor]:
        def net(p):
            return f(p, X)

        if FUSED_SINGLE_COLUMN and not any(_is_transformed(t) for t in v.values()):
            return _ggn_vp_one_pass(net, c, params, loss_args, v)
        pred, Jv = jvp(net, (params,), (v,))
        _, HJv = jvp(jacrev(lambda out: c(out, loss_args)), (pred,), (Jv,))
        _, pull = vjp(net, params)
        (JtHJv,) = pull(HJv)
        return JtHJv

    return ggn_vp


FUSED_SINGLE_COLUMN = True   # (tools/probe_general_direct.py flips it for A/B runs)


def _is_transformed(t: Tensor) -> bool:
    """True inside vmap / jvp / grad of torch.func (the tensor is a functorch wrapper)."""
    try:
        return torch._C._functorch.is_functorch_wrapped_tensor(t)
    except AttributeError:   # pragma: no cover - very old torch
        return False


def _ggn_vp_one_pass(net: Callable, c: Callable, params: dict[str, Tensor], loss_args: tuple,
                     v: dict[str, Tensor]) -> dict[str, Tensor]:
    """``J^T (nabla_f^2 c) J v`` for ONE vector with ONE forward pass: the network runs once on dual numbers (forward-mode
    tangent ``J v``) while the reverse graph of its primal part is recorded, the pull-back then runs on that graph --
    ``jvp`` followed by ``vjp`` (the composition of ``ggn.py:41-72``) evaluates the network twice.  Same arithmetic for the
    prediction, the tangent and the pull-back; only used outside ``vmap`` (single columns)."""
    import torch.autograd.forward_ad as fwAD

    keys = list(params.keys())
    with torch.enable_grad():
        leaves = {k: params[k].detach().requires_grad_(True) for k in keys}
        with fwAD.dual_level():
            out = net({k: fwAD.make_dual(leaves[k], v[k]) for k in keys})
            pred, Jv = fwAD.unpack_dual(out)
        if Jv is None:
            Jv = torch.zeros_like(pred)
        with torch.no_grad():
            _, HJv = jvp(jacrev(lambda o: c(o, loss_args)), (pred.detach(),), (Jv.detach(),))
        if not pred.requires_grad:   # the prediction does not depend on the parameters
            return {k: torch.zeros_like(params[k]) for k in keys}
        grads = torch.autograd.grad(pred, [leaves[k] for k in keys], grad_outputs=HJv, allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(keys, grads)}


def make_batch_hessian_vector_product(f: Callable, loss_func: Module) -> Callable:
    """Forward-over-reverse Hessian-vector product ``jvp(jacrev(loss))`` (``hessian.py:66``)."""
    c = make_functional_loss(loss_func)

    @torch.no_grad()
    def hvp(params, X, loss_args, v):
        _, out = jvp(jacrev(lambda p: c(f(p, X), loss_args)), (params,), (v,))
        return out

    return hvp


def make_batch_ef_vector_product(f: Callable, loss_func: Module) -> Callable:
    """Empirical Fisher as the GGN of ``0.5/c sum_n <f_n, c g_n>^2`` on outputs flattened to
    ``[(batch ...), C]`` (``gradient_moments.py:15-87``)."""
    c = make_functional_loss(loss_func)

    def f_flat(params, X):
        out = f(params, X)
        return out.movedim(1, -1).flatten(0, -2) if isinstance(loss_func, CrossEntropyLoss) else out.flatten(0, -2)

    def c_flat(out_flat, loss_args):
        (y,) = loss_args
        y_flat = y.flatten() if isinstance(loss_func, CrossEntropyLoss) else y.flatten(0, -2)
        return c(out_flat, (y_flat,))

    c_grad = grad(c_flat, argnums=0)

    def pseudo(out_flat, loss_args):
        g = c_grad(out_flat.detach(), loss_args)
        terms, C = out_flat.shape
        red = {"mean": float(terms if isinstance(loss_func, CrossEntropyLoss) else terms * C), "sum": 1.0}[
            loss_func.reduction
        ]
        ip = (out_flat * (g * red)).flatten(1).sum(1)
        return 0.5 / red * (ip**2).sum()

    return make_ggn_vector_product(f_flat, pseudo)


def make_batch_ggn_mc_vector_product(f: Callable, loss_func: Module, mc_samples: int) -> Callable:
    """MC-GGN: GGN of ``0.5/c sum_{n,k} <g'_nk, f_n>^2`` with would-be gradients ``g'`` drawn
    from the model's likelihood by the global RNG (``ggn.py:100-168``)."""
    sampler = vmap(make_grad_output_fn(loss_func, FisherType.MC, mc_samples), (0, 0), randomness="different")

    def pseudo(pred, loss_args):
        (y,) = loss_args
        g = sampler(pred.detach(), y)  # [batch, mc, *out]
        ip = (g * pred.unsqueeze(1)).flatten(2).sum(2)
        red = {"mean": float(pred.shape[0]), "sum": 1.0}[loss_func.reduction]
        return 0.5 / red * (ip**2).sum()

    return make_ggn_vector_product(f, pseudo)


# ------------------------------------------------------------------------------------------
# operators
# ------------------------------------------------------------------------------------------
class HessianLinearOperator(CurvatureLinearOperator):
    """Hessian of the empirical risk, ``c sum_n nabla^2_theta l(f(x_n), y_n)``."""

    SELF_ADJOINT: bool = True
    _NATIVE_KIND = "hessian"

    def _init_mp(self) -> None:
        self._vp = make_batch_hessian_vector_product(self._model_func, self._loss_func)
        super()._init_mp()

    def _matvec_batch(self, X, y, v):
        return self._vp(self._params, X, (y,), v)


class GGNLinearOperator(CurvatureLinearOperator):
    """Generalized Gauss-Newton matrix ``c sum_n J_n^T (nabla_f^2 l_n) J_n``; with
    ``mc_samples > 0`` the loss Hessian is replaced by a Monte-Carlo estimate (seeded per
    product, fixed data order required)."""

    SELF_ADJOINT: bool = True
    MC_SUPPORTED_LOSSES = (MSELoss, CrossEntropyLoss, BCEWithLogitsLoss)
    _NATIVE_KIND = "ggn"

    def __init__(
        self,
        model_func,
        loss_func,
        params: dict[str, Tensor],
        data,
        progressbar: bool = False,
        check_deterministic: bool = True,
        num_data: int | None = None,
        batch_size_fn=None,
        mc_samples: int = 0,
        seed: int = 2147483647,
    ):
        self._mc_samples = mc_samples
        if mc_samples > 0:
            if not isinstance(loss_func, self.MC_SUPPORTED_LOSSES):
                raise NotImplementedError(
                    f"MC-GGN requires loss in {self.MC_SUPPORTED_LOSSES}. Got: {loss_func}."
                )
            self.FIXED_DATA_ORDER = True
            self._seed = seed
            self._NATIVE_KIND = "mc"  # rank-M output curvature from sampled would-be gradients
            self._mc_sampler = vmap(make_grad_output_fn(loss_func, FisherType.MC, mc_samples), (0, 0),
                                    randomness="different")
        super().__init__(
            model_func, loss_func, params, data, progressbar=progressbar,
            check_deterministic=check_deterministic, num_data=num_data, batch_size_fn=batch_size_fn,
        )

    def _init_mp(self) -> None:
        if self._mc_samples > 0:
            self._vp = make_batch_ggn_mc_vector_product(self._model_func, self._loss_func, self._mc_samples)
        else:
            self._vp = make_ggn_vector_product(self._model_func, make_functional_loss(self._loss_func))
        super()._init_mp()

    def _matmat(self, M):
        if self._mc_samples > 0:
            with torch.random.fork_rng():
                torch.manual_seed(self._seed)
                return super()._matmat(M)
        return super()._matmat(M)

    def _matvec_batch(self, X, y, v):
        return self._vp(self._params, X, (y,), v)


class EFLinearOperator(CurvatureLinearOperator):
    """Uncentered gradient covariance ('empirical Fisher'), ``c sum_n g_n g_n^T``."""

    SELF_ADJOINT: bool = True
    SUPPORTED_LOSSES = (MSELoss, CrossEntropyLoss, BCEWithLogitsLoss)
    _NATIVE_KIND = "ef"

    def _init_mp(self) -> None:
        if not isinstance(self._loss_func, self.SUPPORTED_LOSSES):
            raise NotImplementedError(f"Loss must be one of {self.SUPPORTED_LOSSES}. Got: {self._loss_func}.")
        self._vp = make_batch_ef_vector_product(self._model_func, self._loss_func)
        super()._init_mp()

    def _matvec_batch(self, X, y, v):
        return self._vp(self._params, X, (y,), v)


__all__ = [
    "CurvatureLinearOperator", "HessianLinearOperator", "GGNLinearOperator", "EFLinearOperator",
    "make_ggn_vector_product",
]
