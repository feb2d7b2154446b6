"""Thin object wrapper over the C ABI: one ``Engine`` = one ``vp_ctx`` on one MI355X.

This is plumbing (array marshalling, lifetime, fork guard); all arithmetic happens in the HIP
library.  Higher-level mirrors of the reference's interface live in ``rbvfit_amd.model`` and
``rbvfit_amd.vfit``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _lib as L


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def device_count() -> int:
    return int(L.load().vp_device_count())


class Engine:
    """Context on one GPU.  Not fork-safe: the creating pid is recorded and any use from another
    process raises (rbvfit's default ``use_pool=True`` forks; pass ``use_pool=False``)."""

    def __init__(self, device_id: int = 0):
        self._lib = L.load()
        self._ctx = C.c_void_p()
        rc = self._lib.vp_ctx_create(C.byref(self._ctx), int(device_id))
        if rc != L.VP_OK:
            msg = self._lib.vp_last_error(None)
            raise L.RbvfitAmdError(rc, msg.decode() if msg else "vp_ctx_create failed")
        self._pid = os.getpid()
        self.device_id = int(device_id)
        self.ndim = 0
        self.n_pixels = []

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value and os.getpid() == self._pid:
            self._lib.vp_ctx_destroy(self._ctx)
        self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _guard(self):
        if os.getpid() != self._pid:
            raise RuntimeError("rbvfit_amd.Engine used in a forked child: a HIP context does not survive "
                               "fork(); create the engine in the process that uses it (use_pool=False)")
        if not self._ctx.value:
            raise RuntimeError("rbvfit_amd.Engine is closed")

    def _check(self, rc):
        L.check(self._lib, self._ctx, rc)

    # -- setup ------------------------------------------------------------------------------
    def set_option(self, name: str, value: int):
        """Tuning knob of this context (``vp_set_option``): "geom", "finalize", "walker", ...; the
        RBVFIT_AMD_<NAME> environment variables only give the defaults, read when the context is made."""
        self._guard()
        self._check(self._lib.vp_set_option(self._ctx, str(name).encode(), int(value)))

    def set_bounds(self, lb, ub):
        self._guard()
        lb, ub = _f64(lb).ravel(), _f64(ub).ravel()
        if lb.shape != ub.shape:
            raise ValueError("lb and ub must have the same length")
        self._check(self._lib.vp_set_bounds(self._ctx, lb.size, _dp(lb), _dp(ub)))
        self.ndim = lb.size

    def add_instrument(self, wave, flux, inv_sigma2, log_inv_sigma2, lambda0, gamma, f, zfac,
                       N_idx, b_idx, v_idx, taps=None, lsf_mode=L.LSF_NONE, voigt_method=L.VOIGT_WOFZ) -> int:
        self._guard()
        wave, flux, w, lw = _f64(wave), _f64(flux), _f64(inv_sigma2), _f64(log_inv_sigma2)
        if not (wave.ndim == 1 and wave.shape == flux.shape == w.shape == lw.shape):
            raise ValueError("wave, flux, inv_sigma2 and log_inv_sigma2 must be 1-D arrays of equal length")
        lam, gam, fo, zf = _f64(lambda0), _f64(gamma), _f64(f), _f64(zfac)
        ni = np.ascontiguousarray(N_idx, dtype=np.int32)
        bi = np.ascontiguousarray(b_idx, dtype=np.int32)
        vi = np.ascontiguousarray(v_idx, dtype=np.int32)
        if not (lam.shape == gam.shape == fo.shape == zf.shape == ni.shape == bi.shape == vi.shape):
            raise ValueError("line tables must all have shape (n_lines,)")
        if taps is None or lsf_mode == L.LSF_NONE or len(taps) == 0:
            K, tp, lsf_mode = 0, None, L.LSF_NONE
        else:
            t = _f64(taps)
            K, tp = t.size, _dp(t)
        idx = C.c_int(-1)
        self._check(self._lib.vp_add_instrument(
            self._ctx, wave.size, _dp(wave), _dp(flux), _dp(w), _dp(lw), lam.size, _dp(lam), _dp(gam),
            _dp(fo), _dp(zf), _ip(ni), _ip(bi), _ip(vi), K, tp, int(lsf_mode), int(voigt_method), C.byref(idx)))
        self.n_pixels.append(wave.size)
        return idx.value

    def update_spectrum(self, inst, flux, inv_sigma2, log_inv_sigma2):
        self._guard()
        fl, w, lw = _f64(flux), _f64(inv_sigma2), _f64(log_inv_sigma2)
        P = self.n_pixels[inst]
        if not (fl.shape == w.shape == lw.shape == (P,)):
            raise ValueError(f"arrays must have shape ({P},)")
        self._check(self._lib.vp_update_spectrum(self._ctx, int(inst), _dp(fl), _dp(w), _dp(lw)))

    # -- evaluation -------------------------------------------------------------------------
    def _theta2d(self, theta):
        th = _f64(theta)
        if th.ndim == 1:
            th = th[None, :]
        if th.ndim != 2 or th.shape[1] != self.ndim:
            raise ValueError(f"theta must have shape (W, {self.ndim}) or ({self.ndim},); got {np.shape(theta)}")
        return th

    def lnprob(self, theta) -> np.ndarray:
        """(W, D) host array -> (W,) lnprob (H2D, kernels, D2H inside the call).

        This is the seam an ensemble sampler calls once per half-step (vfit_mcmc.py:348-353, 414-439), so the wrapper's own
        cost counts: a C-contiguous float64 (W, D) array goes to the C ABI as it is (no copy, addresses taken through the
        buffer protocol: ~2 us of Python per call instead of ~8 with ``ndarray.ctypes``); anything else is converted first."""
        if os.getpid() != self._pid or not self._ctx.value:
            self._guard()
        th = theta
        if not (type(th) is np.ndarray and th.dtype == np.float64 and th.ndim == 2 and th.flags.c_contiguous
                and th.shape[1] == self.ndim):
            th = self._theta2d(theta)
        W = th.shape[0]
        out = np.empty(W, dtype=np.float64)
        if W == 0:
            self._check(self._lib.vp_lnprob_batch(self._ctx, 0, th.shape[1], None, None))
            return out
        try:
            pt = C.addressof(C.c_char.from_buffer(th))
        except (TypeError, ValueError):          # read-only or otherwise not exportable as a writable buffer
            pt = th.ctypes.data
        rc = self._lib.vp_lnprob_batch(self._ctx, W, th.shape[1], pt, C.addressof(C.c_char.from_buffer(out)))
        if rc:
            self._check(rc)
        return out

    def lnprob_device(self, d_theta_ptr: int, d_out_ptr: int, W: int, stream_ptr: int = 0):
        """Device-resident operands (raw pointers, e.g. ``tensor.data_ptr()``); asynchronous on
        ``stream_ptr`` (a hipStream_t as int).  0 means the CONTEXT's own non-blocking stream -- which is
        also what ``torch.cuda.current_stream().cuda_stream`` yields for torch's default stream, so to
        order the kernels with torch / c10d work make a ``torch.cuda.Stream()`` current and pass its
        handle (``rbvfit_amd.dist`` does)."""
        self._guard()
        self._check(self._lib.vp_lnprob_batch_device(self._ctx, int(W), self.ndim, C.c_void_p(d_theta_ptr),
                                                     C.c_void_p(d_out_ptr), C.c_void_p(stream_ptr)))

    # ---- direct-write gather between the ranks of a multi-process job (vp_gather_*; rbvfit_amd.dist.DirectGather) ----
    def gather_create(self, W: int, world: int, rank: int) -> bytes:
        """This rank's (world, W) gathered vector and flags; returns the 128 bytes of IPC handles its peers need."""
        self._guard()
        buf = C.create_string_buffer(128)
        self._check(self._lib.vp_gather_create(self._ctx, int(W), int(world), int(rank), C.cast(buf, C.c_void_p)))
        return buf.raw

    def gather_connect(self, handles_all: bytes, shared_device: bool = False):
        """Map the peers' vectors: ``handles_all`` = the ranks' handle blocks concatenated in rank order; ``shared_device``:
        some ranks share a GPU (the handshake is then a launch of its own in front of every pass)."""
        self._guard()
        buf = C.create_string_buffer(bytes(handles_all), len(handles_all))
        self._check(self._lib.vp_gather_connect(self._ctx, C.cast(buf, C.c_void_p), int(bool(shared_device))))

    def gather_connect_local(self, peers, shared_device: bool = True):
        """``gather_connect`` for ranks that are ``Engine`` objects of THIS process (``vp_gather_connect_local``): ``peers`` =
        the ranks' engines in rank order (this one's own entry is ignored)."""
        self._guard()
        arr = (C.c_void_p * len(peers))(*[p._ctx.value for p in peers])
        self._check(self._lib.vp_gather_connect_local(self._ctx, arr, int(bool(shared_device))))

    @property
    def device_identity(self) -> str:
        """Host name and PCI bus id of the context's GPU: what tells ranks apart that share a device."""
        import socket
        import torch
        try:
            props = torch.cuda.get_device_properties(self.device_id)
            tag = getattr(props, "uuid", None) or f"{getattr(props, 'pci_bus_id', '')}:{getattr(props, 'pci_device_id', '')}:{self.device_id}"
        except Exception:
            tag = str(self.device_id)
        return f"{socket.gethostname()}/{tag}"

    def lnprob_gather_device(self, d_theta_ptr: int, W: int, stream_ptr: int = 0):
        """One pass: this rank's block evaluated and written into every rank's gathered vector by the kernel itself."""
        self._guard()
        self._check(self._lib.vp_lnprob_gather_device(self._ctx, int(W), self.ndim, C.c_void_p(d_theta_ptr), C.c_void_p(stream_ptr)))

    def gather_wait(self, stream_ptr: int = 0):
        self._guard()
        self._check(self._lib.vp_gather_wait(self._ctx, C.c_void_p(stream_ptr)))

    def gather_state(self):
        """(device pointer of this rank's gathered vector, whether a device-side wait timed out); synchronises."""
        self._guard()
        p, t = C.c_void_p(), C.c_int(0)
        self._check(self._lib.vp_gather_state(self._ctx, C.byref(p), C.byref(t)))
        return int(p.value or 0), bool(t.value)

    def gather_destroy(self):
        self._guard()
        self._check(self._lib.vp_gather_destroy(self._ctx))

    @property
    def stream_handle(self) -> int:
        """hipStream_t of the context's own (non-blocking) stream."""
        self._guard()
        return int(self._lib.vp_ctx_stream(self._ctx) or 0)

    def lnprob_torch(self, theta, out=None):
        """lnprob for a CUDA/HIP ``torch.Tensor`` (W, D) float64 on this engine's GPU, ordered with
        torch's CURRENT stream whatever it is: a non-default stream is used directly; for the default
        stream (handle 0, which the C ABI would read as "the context's stream") the kernels run on
        the context's stream, fenced on both sides with ``wait_stream``.  Returns a (W,) tensor."""
        import torch
        self._guard()
        if theta.dtype != torch.float64 or theta.dim() != 2 or not theta.is_cuda or not theta.is_contiguous():
            raise ValueError("theta must be a contiguous float64 CUDA tensor of shape (W, D)")
        if theta.device.index != self.device_id or theta.shape[1] != self.ndim:
            raise ValueError("theta must live on the engine's GPU and have D = ndim columns")
        W = theta.shape[0]
        if out is None:
            out = torch.empty(W, dtype=torch.float64, device=theta.device)
        cur = torch.cuda.current_stream(theta.device)
        if cur.cuda_stream != 0:
            self.lnprob_device(theta.data_ptr(), out.data_ptr(), W, cur.cuda_stream)
        else:
            own = torch.cuda.ExternalStream(self.stream_handle, device=theta.device)
            own.wait_stream(cur)
            self.lnprob_device(theta.data_ptr(), out.data_ptr(), W, 0)
            cur.wait_stream(own)
        return out

    def model_flux(self, inst: int, theta, convolved: bool = True) -> np.ndarray:
        self._guard()
        th = self._theta2d(theta)
        out = np.empty((th.shape[0], self.n_pixels[inst]), dtype=np.float64)
        self._check(self._lib.vp_model_flux_batch(self._ctx, int(inst), th.shape[0], th.shape[1], _dp(th),
                                                  _dp(out), 1 if convolved else 0))
        return out

    def model_flux_rowsum(self, inst: int, theta, weights, c0: float = 0.0, convolved: bool = True) -> np.ndarray:
        """(W,): c0 - sum_p weights[p] * model_flux(theta_w)[p], reduced on the GPU (``vp_model_flux_rowsum``): the rows stay in HBM."""
        self._guard()
        th = self._theta2d(theta)
        wts = _f64(weights).ravel()
        if wts.size != self.n_pixels[inst]:
            raise ValueError("weights must have one entry per pixel of the instrument")
        out = np.empty(th.shape[0], dtype=np.float64)
        self._check(self._lib.vp_model_flux_rowsum(self._ctx, int(inst), th.shape[0], th.shape[1], _dp(th), _dp(wts), float(c0),
                                                   1 if convolved else 0, _dp(out)))
        return out

    def model_flux_device(self, inst: int, d_theta_ptr: int, d_out_ptr: int, W: int, convolved=True, stream_ptr=0):
        self._guard()
        self._check(self._lib.vp_model_flux_batch_device(self._ctx, int(inst), int(W), self.ndim,
                                                         C.c_void_p(d_theta_ptr), C.c_void_p(d_out_ptr),
                                                         1 if convolved else 0, C.c_void_p(stream_ptr)))

    def model_flux_components(self, inst: int, theta, n_lines: int) -> np.ndarray:
        """(W, L, P): unconvolved exp(-tau_l) of every line (the reference's ``components``)."""
        self._guard()
        th = self._theta2d(theta)
        out = np.empty((th.shape[0], int(n_lines), self.n_pixels[inst]), dtype=np.float64)
        self._check(self._lib.vp_model_flux_components(self._ctx, int(inst), th.shape[0], th.shape[1], _dp(th), _dp(out)))
        return out

    def voigt_h(self, a, x) -> np.ndarray:
        """H(a_i, x_j) grid on the device (test hook for the Faddeeva tiers)."""
        self._guard()
        a, x = _f64(a).ravel(), _f64(x).ravel()
        out = np.empty((a.size, x.size), dtype=np.float64)
        self._check(self._lib.vp_voigt_h(self._ctx, a.size, _dp(a), x.size, _dp(x), _dp(out)))
        return out

    # -- device-resident ensemble sampler (vp_stretch_run) ---------------------------------------
    def stretch_run(self, pos, nsteps: int, lnprob=None, a: float = 2.0, seed: int = 0, step0: int = 0,
                    store_chain: bool = True, naccepted=None):
        """``nsteps`` stretch-move iterations on the GPU (positions, lnprob, proposals and
        accept/reject stay in HBM).  Returns (pos, lnprob, chain, chain_lnprob, naccepted);
        chain arrays are None when ``store_chain`` is False.  Raises ValueError when a proposal's
        lnprob is NaN, as emcee does."""
        self._guard()
        pos = np.array(pos, dtype=np.float64, order="C")
        if pos.ndim != 2:
            raise ValueError("pos must have shape (nwalkers, ndim)")
        W, D = pos.shape
        have = lnprob is not None
        lp = np.array(lnprob, dtype=np.float64) if have else np.empty(W, dtype=np.float64)
        if lp.shape != (W,):
            raise ValueError("lnprob must have shape (nwalkers,)")
        chain = np.empty((nsteps, W, D), dtype=np.float64) if store_chain else None
        clp = np.empty((nsteps, W), dtype=np.float64) if store_chain else None
        nacc = np.zeros(W, dtype=np.int64) if naccepted is None else np.ascontiguousarray(naccepted, dtype=np.int64)
        rc = self._lib.vp_stretch_run(self._ctx, W, D, _dp(pos), _dp(lp), 1 if have else 0, int(nsteps), float(a),
                                      C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint64(int(step0)),
                                      _dp(chain) if store_chain else None, _dp(clp) if store_chain else None,
                                      nacc.ctypes.data_as(C.POINTER(C.c_int64)))
        if rc == L.VP_ENAN:
            raise ValueError("Probability function returned NaN")
        self._check(rc)
        return pos, lp, chain, clp, nacc

    # -- device-resident ensemble slice sampler (vp_slice_run) ------------------------------------
    def slice_run(self, pos, nsteps: int, lnprob=None, mu: float = 1.0, tune: bool = True, tolerance: float = 0.05,
                  patience: int = 5, maxsteps: int = 10000, seed: int = 0, step0: int = 0, store_chain: bool = True):
        """``nsteps`` iterations of ensemble slice sampling (zeus' differential move) on the GPU; the ragged
        active sets of the stepping-out / shrinking rounds are compacted in HBM.  Returns a dict: pos, lnprob,
        chain, chain_lnprob (None when ``store_chain`` is False), mu, tune (still adapting?), tune_state (pass it back as
        ``tune`` to continue the run: the patience counter travels with it), mu_history, n_evals."""
        self._guard()
        pos = np.array(pos, dtype=np.float64, order="C")
        if pos.ndim != 2:
            raise ValueError("pos must have shape (nwalkers, ndim)")
        W, D = pos.shape
        have = lnprob is not None
        lp = np.array(lnprob, dtype=np.float64) if have else np.empty(W, dtype=np.float64)
        if lp.shape != (W,):
            raise ValueError("lnprob must have shape (nwalkers,)")
        chain = np.empty((nsteps, W, D), dtype=np.float64) if store_chain else None
        clp = np.empty((nsteps, W), dtype=np.float64) if store_chain else None
        hist = np.empty(max(nsteps, 1), dtype=np.float64)
        # tune: False / True, or the integer state a previous call returned in "tune_state" (1 + consecutive in-tolerance iterations)
        c_mu, c_tune, c_ne = C.c_double(float(mu)), C.c_int(int(tune) if tune else 0), C.c_int64(0)
        rc = self._lib.vp_slice_run(self._ctx, W, D, _dp(pos), _dp(lp), 1 if have else 0, int(nsteps), C.byref(c_mu),
                                    C.byref(c_tune), float(tolerance), int(patience), int(maxsteps),
                                    C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint64(int(step0)),
                                    _dp(chain) if store_chain else None, _dp(clp) if store_chain else None, _dp(hist),
                                    C.byref(c_ne))
        if rc == L.VP_ENAN:
            msg = self._lib.vp_last_error(self._ctx)
            raise ValueError(msg.decode() if msg else "Log Probability returned NaN")
        self._check(rc)
        return dict(pos=pos, lnprob=lp, chain=chain, chain_lnprob=clp, mu=c_mu.value, tune=bool(c_tune.value),
                    tune_state=int(c_tune.value), mu_history=hist[:nsteps], n_evals=int(c_ne.value))

    @property
    def last_launch_kind(self) -> str:
        """'walker' when the last lnprob batch ran as ONE walker_kernel launch, 'tiles' for prep + tile (+ finalize)
        launches, 'tiles+farfield' when those took far lines from per-block expansions (farfield_kernel), 'tiles-multi' when the
        tiles of several instruments that share their records ran as one launch (tile_kernel_multi)."""
        self._guard()
        kind = self._lib.vp_last_launch_kind(self._ctx)
        return {1: "walker", 2: "tiles+farfield", 3: "tiles-multi"}.get(kind, "tiles")

    @property
    def last_walker_split(self) -> int:
        """Workgroups per walker of the last walker_kernel launch (``vp_last_walker_split``): 0 the ordinary form, 2 / 4 / 8 its
        split form for batches of at most one walker per compute unit (option "walker_split")."""
        self._guard()
        return int(self._lib.vp_last_walker_split(self._ctx))

    @property
    def last_farfield_info(self) -> dict:
        """What the far-field expansions of the last lnprob batch covered (``vp_last_farfield_info``): ``variant`` 'none',
        'lines+clusters' (farfield_kernel<6,false>) or 'members' (farfield_kernel<9,true>: narrow-pixel instruments, members
        of near clusters line by line); ``covered`` / ``covered_members`` (walker, block, line) triples; ``pairs`` all triples."""
        self._guard()
        v, a, b, n = C.c_int(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(self._lib.vp_last_farfield_info(self._ctx, C.byref(v), C.byref(a), C.byref(b), C.byref(n)))
        return dict(variant={0: "none", 1: "lines+clusters", 2: "members"}[v.value], covered=a.value,
                    covered_members=b.value, pairs=n.value)

    @property
    def prearm_counts(self) -> dict:
        """Pre-armed launches of the host-buffer lnprob entry (``vp_prearm_counts``; option "prearm"): calls started through
        one (``used``), launches that gave up waiting (``expired``) and launches sent away unused (``cancelled``)."""
        self._guard()
        u, e, x = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(self._lib.vp_prearm_counts(self._ctx, C.byref(u), C.byref(e), C.byref(x)))
        return dict(used=u.value, expired=e.value, cancelled=x.value)

    # -- per-kernel timing (HIP events on the launch stream) ------------------------------------
    def profile_enable(self, on: bool = True):
        self._guard()
        self._check(self._lib.vp_profile_enable(self._ctx, 1 if on else 0))

    def profile_read(self):
        """-> dict(prep_ms, tile_ms, finalize_ms, n_tile_launches) summed since the last read."""
        self._guard()
        a, b, c_, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        self._check(self._lib.vp_profile_read(self._ctx, C.byref(a), C.byref(b), C.byref(c_), C.byref(n)))
        return dict(prep_ms=a.value, tile_ms=b.value, finalize_ms=c_.value, n_tile_launches=n.value)


class MultiEngine:
    """Several GPUs from ONE process through ``vp_multi_*`` (no torch, no RCCL): one context per entry of
    ``device_ids`` (a device may appear more than once), identical static data on each, and ``lnprob``
    shards the walker rows in contiguous blocks of ceil(W / n) -- ``rbvfit_amd.dist.shard_bounds`` --
    with every block in flight before any is waited for.  The torch-free counterpart of
    ``rbvfit_amd.dist.ShardedPosterior`` (which is one process per GPU + an RCCL all-gather)."""

    def __init__(self, device_ids: Sequence[int]):
        self._lib = L.load()
        self._m = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        rc = self._lib.vp_multi_create(C.byref(self._m), len(device_ids), ids)
        if rc != L.VP_OK:
            msg = self._lib.vp_multi_last_error(None)
            raise L.RbvfitAmdError(rc, msg.decode() if msg else "vp_multi_create failed")
        self._pid = os.getpid()
        self.device_ids = [int(d) for d in device_ids]
        self.ndim = 0

    def close(self):
        if getattr(self, "_m", None) and self._m.value and os.getpid() == self._pid:
            self._lib.vp_multi_destroy(self._m)
        self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _guard(self):
        if os.getpid() != self._pid:
            raise RuntimeError("rbvfit_amd.MultiEngine used in a forked child (a HIP context does not survive fork())")
        if not self._m.value:
            raise RuntimeError("rbvfit_amd.MultiEngine is closed")

    def _check(self, rc):
        if rc != L.VP_OK:
            msg = self._lib.vp_multi_last_error(self._m)
            raise L.RbvfitAmdError(rc, msg.decode() if msg else "unknown error")

    @property
    def n_devices(self) -> int:
        return int(self._lib.vp_multi_n_devices(self._m))

    def set_option(self, name: str, value: int):
        self._guard()
        for i in range(self.n_devices):
            ctx = C.c_void_p(self._lib.vp_multi_ctx(self._m, i))
            L.check(self._lib, ctx, self._lib.vp_set_option(ctx, str(name).encode(), int(value)))

    def set_bounds(self, lb, ub):
        self._guard()
        lb, ub = _f64(lb).ravel(), _f64(ub).ravel()
        if lb.shape != ub.shape:
            raise ValueError("lb and ub must have the same length")
        self._check(self._lib.vp_multi_set_bounds(self._m, lb.size, _dp(lb), _dp(ub)))
        self.ndim = lb.size

    def add_instrument(self, wave, flux, inv_sigma2, log_inv_sigma2, lambda0, gamma, f, zfac,
                       N_idx, b_idx, v_idx, taps=None, lsf_mode=L.LSF_NONE, voigt_method=L.VOIGT_WOFZ) -> int:
        self._guard()
        wave, flux, w, lw = _f64(wave), _f64(flux), _f64(inv_sigma2), _f64(log_inv_sigma2)
        if not (wave.ndim == 1 and wave.shape == flux.shape == w.shape == lw.shape):
            raise ValueError("wave, flux, inv_sigma2 and log_inv_sigma2 must be 1-D arrays of equal length")
        lam, gam, fo, zf = _f64(lambda0), _f64(gamma), _f64(f), _f64(zfac)
        ni = np.ascontiguousarray(N_idx, dtype=np.int32)
        bi = np.ascontiguousarray(b_idx, dtype=np.int32)
        vi = np.ascontiguousarray(v_idx, dtype=np.int32)
        if taps is None or lsf_mode == L.LSF_NONE or len(taps) == 0:
            K, tp, lsf_mode = 0, None, L.LSF_NONE
        else:
            t = _f64(taps)
            K, tp = t.size, _dp(t)
        idx = C.c_int(-1)
        self._check(self._lib.vp_multi_add_instrument(
            self._m, wave.size, _dp(wave), _dp(flux), _dp(w), _dp(lw), lam.size, _dp(lam), _dp(gam),
            _dp(fo), _dp(zf), _ip(ni), _ip(bi), _ip(vi), K, tp, int(lsf_mode), int(voigt_method), C.byref(idx)))
        return idx.value

    def lnprob(self, theta) -> np.ndarray:
        self._guard()
        th = _f64(theta)
        if th.ndim == 1:
            th = th[None, :]
        if th.ndim != 2 or th.shape[1] != self.ndim:
            raise ValueError(f"theta must have shape (W, {self.ndim}) or ({self.ndim},); got {np.shape(theta)}")
        out = np.empty(th.shape[0], dtype=np.float64)
        self._check(self._lib.vp_multi_lnprob_batch(self._m, th.shape[0], th.shape[1], _dp(th), _dp(out)))
        return out

    def stretch_run(self, pos, nsteps: int, lnprob=None, a: float = 2.0, seed: int = 0, step0: int = 0,
                    store_chain: bool = True, naccepted=None):
        """``Engine.stretch_run`` for ONE ensemble sharded over this object's device contexts (``vp_multi_stretch_run``):
        every context keeps the whole ensemble in HBM and runs its block of each half-step; moved rows are written into
        every replica, half-steps are ordered by events.  Same arguments and results -- the same chain, bit for bit,
        whatever the number of contexts."""
        self._guard()
        pos = np.array(pos, dtype=np.float64, order="C")
        if pos.ndim != 2:
            raise ValueError("pos must have shape (nwalkers, ndim)")
        W, D = pos.shape
        have = lnprob is not None
        lp = np.array(lnprob, dtype=np.float64) if have else np.empty(W, dtype=np.float64)
        if lp.shape != (W,):
            raise ValueError("lnprob must have shape (nwalkers,)")
        chain = np.empty((nsteps, W, D), dtype=np.float64) if store_chain else None
        clp = np.empty((nsteps, W), dtype=np.float64) if store_chain else None
        nacc = np.zeros(W, dtype=np.int64) if naccepted is None else np.ascontiguousarray(naccepted, dtype=np.int64)
        rc = self._lib.vp_multi_stretch_run(self._m, W, D, _dp(pos), _dp(lp), 1 if have else 0, int(nsteps), float(a),
                                            C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint64(int(step0)),
                                            _dp(chain) if store_chain else None, _dp(clp) if store_chain else None,
                                            nacc.ctypes.data_as(C.POINTER(C.c_int64)))
        if rc == L.VP_ENAN:
            raise ValueError("Probability function returned NaN")
        self._check(rc)
        return pos, lp, chain, clp, nacc

    def slice_run(self, pos, nsteps: int, lnprob=None, mu: float = 1.0, tune=True, tolerance: float = 0.05,
                  patience: int = 5, maxsteps: int = 10000, seed: int = 0, step0: int = 0, store_chain: bool = True):
        """``Engine.slice_run`` for ONE ensemble on this object's device contexts (``vp_multi_slice_run``): the sampler
        state is replicated, every round's lnprob batch is cut into one block of trial rows per context.  Same arguments
        and results -- the same chain, bit for bit, whatever the number of contexts."""
        self._guard()
        pos = np.array(pos, dtype=np.float64, order="C")
        if pos.ndim != 2:
            raise ValueError("pos must have shape (nwalkers, ndim)")
        W, D = pos.shape
        have = lnprob is not None
        lp = np.array(lnprob, dtype=np.float64) if have else np.empty(W, dtype=np.float64)
        if lp.shape != (W,):
            raise ValueError("lnprob must have shape (nwalkers,)")
        chain = np.empty((nsteps, W, D), dtype=np.float64) if store_chain else None
        clp = np.empty((nsteps, W), dtype=np.float64) if store_chain else None
        hist = np.empty(max(nsteps, 1), dtype=np.float64)
        c_mu, c_tune, c_ne = C.c_double(float(mu)), C.c_int(int(tune) if tune else 0), C.c_int64(0)
        rc = self._lib.vp_multi_slice_run(self._m, W, D, _dp(pos), _dp(lp), 1 if have else 0, int(nsteps), C.byref(c_mu),
                                          C.byref(c_tune), float(tolerance), int(patience), int(maxsteps),
                                          C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint64(int(step0)),
                                          _dp(chain) if store_chain else None, _dp(clp) if store_chain else None, _dp(hist),
                                          C.byref(c_ne))
        if rc == L.VP_ENAN:
            msg = self._lib.vp_multi_last_error(self._m)
            raise ValueError(msg.decode() if msg else "Log Probability returned NaN")
        self._check(rc)
        return dict(pos=pos, lnprob=lp, chain=chain, chain_lnprob=clp, mu=c_mu.value, tune=bool(c_tune.value),
                    tune_state=int(c_tune.value), mu_history=hist[:nsteps], n_evals=int(c_ne.value))
