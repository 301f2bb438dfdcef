"""Python host mirror of the AeroBulk public interface, on top of the C ABI.

Names and argument meaning follow the reference's Fortran API
(`AEROBULK_MODEL`, src/mod_aerobulk.f90:176-230) so that parity tests read like the
reference's own example drivers (src/tests/example_call_aerobulk.f90:48-51).

Arrays may be numpy float64 arrays (host: the library stages them through HBM) or torch CUDA
tensors (device-resident: zero-copy, asynchronous).  PyTorch is used only as a device-memory
and stream provider.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

ALGOS = {"coare3p0": 1, "coare3p6": 2, "ncar": 3, "ecmwf": 4, "andreas": 5}
HUM_TYPES = {0: "sh", 1: "dp", 2: "rh"}
HUM_IDS = {v: k for k, v in HUM_TYPES.items()}
AB_MEM_HOST, AB_MEM_DEVICE = 0, 1
AB_F64, AB_F32, AB_F32_STORAGE, AB_F32_MIXED = 0, 1, 2, 3
PRECISIONS = {"f64": AB_F64, "f32": AB_F32, "f32_storage": AB_F32_STORAGE, "f32_mixed": AB_F32_MIXED}   # f32_storage: fp32 arrays, fp64 arithmetic; f32_mixed: fp32 arrays, fp64 anchors
AB_ERR_TAU = 8
IN_NAMES = ("sst", "t_zt", "hum_zt", "U_zu", "V_zu", "slp", "rad_sw", "rad_lw")
OUT_NAMES = ("QL", "QH", "Tau_x", "Tau_y", "Evap", "T_s")


class AerobulkError(RuntimeError):
    """A condition on which the reference prints a message and STOPs (mod_const.f90:238-278)."""

    def __init__(self, status, message):
        super().__init__(f"[ab_status {status}] {message}")
        self.status = status
        self.message = message


def _raise(status):
    lib = _lib.load()
    msg = lib.ab_last_error().decode(errors="replace") or lib.ab_strerror(status).decode()
    raise AerobulkError(status, msg)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x, dtype, n):
    """(address, keepalive) of a flat field; None -> NULL."""
    if x is None:
        return None, None
    if _is_torch(x):
        if not x.is_cuda:
            raise ValueError("torch tensors must live on the GPU (use numpy arrays for host data)")
        import torch
        want = torch.float64 if dtype == np.float64 else torch.float32
        if x.dtype != want or not x.is_contiguous():
            raise ValueError(f"device field must be contiguous {want}")
        if x.numel() != n:
            raise ValueError(f"field has {x.numel()} cells, expected {n}")
        return x.data_ptr(), x
    a = np.ascontiguousarray(x, dtype=dtype)
    if a.size != n:
        raise ValueError(f"field has {a.size} cells, expected {n}")
    return a.ctypes.data, a


def _stream_of(x, stream):
    """hipStream_t for device fields: the caller's, else torch's current stream (the one their producers ran on)."""
    if stream is None and x is not None and _is_torch(x):
        import torch
        stream = torch.cuda.current_stream(x.device).cuda_stream
    return C.c_void_p(stream or 0)


class Session:
    """One aerobulk_model() time loop (jt = 1..Nt) on one MI355X, or sharded by row blocks over several.

    Replaces the reference's module-global state by an explicit handle.  `device`: a HIP ordinal (-1 = current),
    "all" (one j-block per visible GPU) or a list of ordinals (one j-block each; an ordinal may repeat).  `rows`: with a list of
    ordinals, the number of rows of each j-block (default: equal blocks)."""

    def __init__(self, calgo, Ni, Nj=1, Nt=1, l_use_skin=False, precision="f64", device=-1, rows=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        algo = self._lib.ab_algo_from_string(calgo.encode(), -1)
        if algo == 0:
            raise AerobulkError(1, f"ERROR: mod_aerobulk_compute.f90 => bulk algorithm {calgo} is unknown!!!")
        self.calgo, self.Ni, self.Nj, self.Nt = calgo, int(Ni), int(Nj), int(Nt)
        self.n = self.Ni * self.Nj
        self.l_use_skin = bool(l_use_skin)
        self.dtype = np.float64 if precision == "f64" else np.float32
        prec = PRECISIONS[precision]
        if isinstance(device, (list, tuple)):
            devs = (C.c_int * len(device))(*[int(d) for d in device])
            if rows is not None:
                if len(rows) != len(device):
                    raise ValueError("rows: one count per shard")
                nrows = (C.c_long * len(device))(*[int(r) for r in rows])
                rc = self._lib.ab_session_create_sharded_rows(C.byref(self._h), algo, self.Ni, self.Nj, self.Nt, int(self.l_use_skin),
                                                              prec, devs, len(device), nrows)
            else:
                rc = self._lib.ab_session_create_sharded(C.byref(self._h), algo, self.Ni, self.Nj, self.Nt, int(self.l_use_skin),
                                                         prec, devs, len(device))
        else:
            rc = self._lib.ab_session_create(C.byref(self._h), algo, self.Ni, self.Nj, self.Nt, int(self.l_use_skin), prec,
                                             -2 if device == "all" else int(device))
        if rc:
            _raise(rc)
        self.hum_type = "sh"

    def shards(self):
        """[(j0, nj_local, device)] of the row blocks (one entry for an ordinary session)."""
        out = []
        for r in range(self._lib.ab_session_shard_count(self._h)):
            j0, njl, dev = C.c_long(), C.c_long(), C.c_int()
            rc = self._lib.ab_session_shard_info(self._h, r, C.byref(j0), C.byref(njl), C.byref(dev))
            if rc:
                _raise(rc)
            out.append((j0.value, njl.value, dev.value))
        return out

    # -- AEROBULK_INIT (mod_aerobulk.f90:24-160)
    def init(self, sst, t_zt, hum_zt, U_zu, V_zu, slp, rad_sw=None, rad_lw=None, stream=None):
        fields = [sst, t_zt, hum_zt, U_zu, V_zu, slp, rad_sw, rad_lw]
        dev = _is_torch(sst)
        ptrs, keep = zip(*[_ptr(f, self.dtype, self.n) for f in fields])
        rep = _lib.InitReport()
        rc = self._lib.ab_session_init(self._h, *ptrs, AB_MEM_DEVICE if dev else AB_MEM_HOST, _stream_of(sst, stream), C.byref(rep))
        report = dict(n_cells=rep.n_cells, n_masked=rep.n_masked, hum_type=HUM_TYPES.get(rep.hum_type),
                      bad_field=rep.bad_field, bad_min=rep.bad_min, bad_max=rep.bad_max, bad_mean=rep.bad_mean)
        if rc:
            _raise(rc)
        self.hum_type = report["hum_type"]
        return report

    # -- AEROBULK_INIT for a sharded grid: local statistics, then decisions on the combined statistics
    def init_stats(self, sst, t_zt, hum_zt, U_zu, V_zu, slp, rad_sw=None, rad_lw=None, stream=None):
        """29 doubles: [0:11] combine by SUM, [11:20] by MIN, [20:29] by MAX (include/aerobulk_amd.h)."""
        fields = [sst, t_zt, hum_zt, U_zu, V_zu, slp, rad_sw, rad_lw]
        ptrs, keep = zip(*[_ptr(f, self.dtype, self.n) for f in fields])
        st = np.empty(29)
        rc = self._lib.ab_session_init_stats(self._h, *ptrs, AB_MEM_DEVICE if _is_torch(sst) else AB_MEM_HOST,
                                             _stream_of(sst, stream), st.ctypes.data_as(_lib.dp))
        if rc:
            _raise(rc)
        return st

    def init_apply(self, stats, have_rad=False):
        st = np.ascontiguousarray(stats, dtype=np.float64)
        rep = _lib.InitReport()
        rc = self._lib.ab_session_init_apply(self._h, st.ctypes.data_as(_lib.dp), int(have_rad), C.byref(rep))
        report = dict(n_cells=rep.n_cells, n_masked=rep.n_masked, hum_type=HUM_TYPES.get(rep.hum_type),
                      bad_field=rep.bad_field, bad_min=rep.bad_min, bad_max=rep.bad_max, bad_mean=rep.bad_mean)
        if rc:
            _raise(rc)
        self.hum_type = report["hum_type"]
        return report

    def set_humidity(self, hum_type):
        rc = self._lib.ab_session_set_humidity(self._h, HUM_IDS[hum_type])
        if rc:
            _raise(rc)
        self.hum_type = hum_type

    def set_diagnostics(self, names=None, device=None):
        """Ask every following compute() for the TURB_* diagnostics in `names` (any of _lib.Diag.NAMES; None/empty switches
        them off).  Returns the dict of arrays that will be (re)written: numpy arrays, or torch tensors on `device`."""
        self._diag = {}
        if not names:
            rc = self._lib.ab_session_set_diagnostics(self._h, None, 0)
            if rc:
                _raise(rc)
            return self._diag
        d = _lib.Diag()
        for k in names:
            if k not in _lib.Diag.NAMES:
                raise ValueError(f"unknown diagnostic {k}")
            if device is not None:
                import torch
                a = torch.empty(self.n, dtype=torch.float64 if self.dtype == np.float64 else torch.float32, device=device)
                setattr(d, k, a.data_ptr())
            else:
                a = np.empty(self.n, dtype=self.dtype)
                setattr(d, k, a.ctypes.data)
            self._diag[k] = a
        rc = self._lib.ab_session_set_diagnostics(self._h, C.byref(d), AB_MEM_DEVICE if device is not None else AB_MEM_HOST)
        if rc:
            _raise(rc)
        return self._diag

    def set_regroup(self, on=True):
        """Lane regrouping of the flux kernel (ab_session_set_regroup): results do not depend on it."""
        rc = self._lib.ab_session_set_regroup(self._h, int(bool(on)))
        if rc:
            _raise(rc)

    def set_solar_time(self, isecday_utc, lon=None, stream=None):
        p, keep = _ptr(lon, self.dtype, self.n)
        rc = self._lib.ab_session_set_solar_time(self._h, int(isecday_utc), p,
                                                 AB_MEM_DEVICE if (lon is not None and _is_torch(lon)) else AB_MEM_HOST,
                                                 _stream_of(lon, stream))
        if rc:
            _raise(rc)

    # -- aerobulk_compute (mod_aerobulk_compute.f90:22-213)
    def compute(self, jt, zt, zu, sst, t_zt, hum_zt, U_zu, V_zu, slp, Niter=5, rad_sw=None, rad_lw=None,
                out=None, want_T_s=None, stream=None, check=True):
        dev = _is_torch(sst)
        ins = [sst, t_zt, hum_zt, U_zu, V_zu, slp, rad_sw, rad_lw]
        iptr, ikeep = zip(*[_ptr(f, self.dtype, self.n) for f in ins])
        if want_T_s is None:
            want_T_s = rad_sw is not None and rad_lw is not None
        names = OUT_NAMES if want_T_s else OUT_NAMES[:5]
        if out is None:
            if dev:
                import torch
                out = {k: torch.empty(self.n, dtype=sst.dtype, device=sst.device) for k in names}
            else:
                out = {k: np.empty(self.n, dtype=self.dtype) for k in names}
        optr = []
        for k in OUT_NAMES:
            o = out.get(k)
            optr.append(_ptr(o, self.dtype, self.n)[0] if o is not None else None)
            if o is not None and not dev and not (isinstance(o, np.ndarray) and o.flags["C_CONTIGUOUS"] and o.dtype == self.dtype):
                raise ValueError("host outputs must be contiguous arrays of the session precision")
        if dev and stream is None:
            import torch
            stream = torch.cuda.current_stream().cuda_stream
        rc = self._lib.ab_session_compute(self._h, int(jt), float(zt), float(zu), int(Niter), *iptr, *optr,
                                          AB_MEM_DEVICE if dev else AB_MEM_HOST, C.c_void_p(stream or 0))
        if rc:
            _raise(rc)
        if dev and check:
            self.check()
        return out

    # -- device-resident fields of a sharded session, and the gather of the fluxes (the north_star's multi-GPU layout)
    def _shard_structs(self, shard_fields, shard_out):
        nsh = self._lib.ab_session_shard_count(self._h)
        if len(shard_fields) != nsh or len(shard_out) != nsh:
            raise ValueError(f"{nsh} shards: one dict of fields / outputs per shard")
        arr = (_lib.ShardArrays * nsh)()
        keep = []
        for r, (j0, njl, dev) in enumerate(self.shards()):
            n = self.Ni * njl
            f, o = shard_fields[r], shard_out[r]
            for name, key in (("sst", "sst"), ("t_zt", "t_zt"), ("hum_zt", "hum_zt"), ("u_zu", "U_zu"), ("v_zu", "V_zu"), ("slp", "slp"),
                              ("rad_sw", "rad_sw"), ("rad_lw", "rad_lw")):
                x = f.get(key)
                if x is not None:
                    ptr, k = _ptr(x, self.dtype, n)
                    keep.append(k)
                    setattr(arr[r], name, ptr.value if hasattr(ptr, "value") else ptr)
            for name, key in (("ql", "QL"), ("qh", "QH"), ("tau_x", "Tau_x"), ("tau_y", "Tau_y"), ("evap", "Evap"), ("t_s", "T_s")):
                x = o.get(key)
                if x is not None:
                    ptr, k = _ptr(x, self.dtype, n)
                    keep.append(k)
                    setattr(arr[r], name, ptr.value if hasattr(ptr, "value") else ptr)
        return arr, keep

    def compute_shards(self, jt, zt, zu, shard_fields, shard_out, Niter=5, streams=None, check=True):
        """aerobulk_compute on device-resident fields, ONE DICT OF TORCH TENSORS PER SHARD (keys as compute(): sst t_zt hum_zt U_zu V_zu slp
        [rad_sw rad_lw]; outputs QL QH Tau_x Tau_y [Evap T_s]), each holding that shard's rows on that shard's device."""
        arr, keep = self._shard_structs(shard_fields, shard_out)
        st = None
        if streams is not None:
            st = (C.c_void_p * len(streams))(*[C.c_void_p(x or 0) for x in streams])
        rc = self._lib.ab_session_compute_shards(self._h, int(jt), float(zt), float(zu), int(Niter), arr, st)
        if rc:
            _raise(rc)
        if check:
            self.check()

    def gather(self, shard_out, dst, root=0, streams=None, synchronize=True):
        """The fluxes of every shard into whole-grid tensors `dst` (dict: QL QH Tau_x Tau_y [Evap T_s]; missing = not gathered) on the
        device of shard `root`: RCCL send / recv inside the process for shards on other devices, device-to-device copies otherwise."""
        nsh = self._lib.ab_session_shard_count(self._h)
        arr, keep = self._shard_structs([{}] * nsh, shard_out)
        d = _lib.FluxArrays()
        for name, key in (("ql", "QL"), ("qh", "QH"), ("tau_x", "Tau_x"), ("tau_y", "Tau_y"), ("evap", "Evap"), ("t_s", "T_s")):
            x = dst.get(key)
            if x is not None:
                ptr, k = _ptr(x, self.dtype, self.n)
                keep.append(k)
                setattr(d, name, ptr.value if hasattr(ptr, "value") else ptr)
        st = None
        if streams is not None:
            st = (C.c_void_p * len(streams))(*[C.c_void_p(x or 0) for x in streams])
        rc = self._lib.ab_session_gather(self._h, int(root), arr, C.byref(d), st, int(bool(synchronize)))
        if rc:
            _raise(rc)

    # -- TURB_<algo> itself (mod_blk_coare3p6.f90:123-131 and siblings)
    def turb(self, kt, zt, zu, T_s, theta_zt, q_s, q_zt, U_zu, l_use_cs=False, l_use_wl=False, Qsw=None, rad_lw=None, slp=None,
             nb_iter=5, stream=None):
        """One call of the session algorithm's TURB_* routine.  `T_s` and `q_s` are updated IN PLACE when a skin scheme is on
        (INTENT(inout) in the reference); returns the dict of the mandatory outputs Cd Ch Ce t_zu q_zu Ubzu.  OPTIONAL
        outputs: set_diagnostics().  Arrays: numpy (host) or torch (device), flat, session precision."""
        dev = _is_torch(T_s)
        for a in (T_s, q_s):
            if not dev and not (isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"] and a.dtype == self.dtype):
                raise ValueError("T_s / q_s are updated in place: contiguous arrays of the session precision required")
        f = _lib.TurbFields()
        keep = []
        for k, a in (("T_s", T_s), ("theta_zt", theta_zt), ("q_s", q_s), ("q_zt", q_zt), ("U_zu", U_zu), ("Qsw", Qsw),
                     ("rad_lw", rad_lw), ("slp", slp)):
            p, kp = _ptr(a, self.dtype, self.n)
            keep.append(kp)
            setattr(f, k, p)
        if dev:
            import torch
            out = {k: torch.empty(self.n, dtype=T_s.dtype, device=T_s.device) for k in _lib.Diag.NAMES[:6]}
            if stream is None:
                stream = torch.cuda.current_stream().cuda_stream
        else:
            out = {k: np.empty(self.n, dtype=self.dtype) for k in _lib.Diag.NAMES[:6]}
        for k, a in out.items():
            setattr(f, k, _ptr(a, self.dtype, self.n)[0])
        rc = self._lib.ab_session_turb(self._h, int(kt), float(zt), float(zu), int(bool(l_use_cs)), int(bool(l_use_wl)),
                                       int(nb_iter), C.byref(f), AB_MEM_DEVICE if dev else AB_MEM_HOST, C.c_void_p(stream or 0))
        if rc:
            _raise(rc)
        if dev:
            import torch
            torch.cuda.synchronize()
        return out

    def check(self):
        rc = self._lib.ab_session_check(self._h)
        if rc:
            _raise(rc)

    def wl_state(self):
        st = np.empty(4 * self.n)
        rc = self._lib.ab_session_get_wl_state(self._h, st.ctypes.data_as(_lib.dp))
        if rc:
            _raise(rc)
        return dict(zip(("dT_wl", "Hz_wl", "Qnt_ac", "Tau_ac"), st.reshape(4, self.n)))

    def last_kernel_ms(self):
        return self._lib.ab_session_last_kernel_ms(self._h)

    def close(self):
        if self._h:
            self._lib.ab_session_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def aerobulk_model(jt, Nt, calgo, zt, zu, sst, t_zt, hum_zt, U_zu, V_zu, slp, Niter=None, l_use_skin=False,
                   rad_sw=None, rad_lw=None):
    """AEROBULK_MODEL (src/mod_aerobulk.f90:176-269) through the library's process-global session:
    INIT checks at jt==1, sticky Niter, non-reentrant — the reference's own protocol.
    Host float64 arrays of any shape (flattened in Fortran order is the caller's business: cells
    are independent).  Returns dict(QL, QH, Tau_x, Tau_y, Evap[, T_s]) shaped like `sst`."""
    lib = _lib.load()
    shape = np.shape(sst)
    n = int(np.prod(shape))
    ni = shape[0] if len(shape) > 0 else 1
    nj = n // ni
    ins = [np.ascontiguousarray(np.ravel(f, order="F"), dtype=np.float64) for f in (sst, t_zt, hum_zt, U_zu, V_zu, slp)]
    for a in ins:
        if a.size != n:  # mod_aerobulk.f90:87-91
            raise AerobulkError(10, " AEROBULK_INIT => SST and input arrays do not agree in shape!")
    rad = [None, None]
    if rad_sw is not None and rad_lw is not None:
        rad = [np.ascontiguousarray(np.ravel(f, order="F"), dtype=np.float64) for f in (rad_sw, rad_lw)]
        for a in rad:
            if a.size != n:  # :93-94
                raise AerobulkError(10, " AEROBULK_INIT => SST and Rad arrays do not agree in shape!")
    outs = [np.empty(n) for _ in range(5)]
    t_s = np.empty(n) if rad[0] is not None else None
    P = lambda a: a.ctypes.data_as(_lib.dp) if a is not None else C.cast(None, _lib.dp)
    rep = _lib.InitReport()
    rc = lib.ab_model(int(jt), int(Nt), calgo.encode(), len(calgo), float(zt), float(zu), *[P(a) for a in ins],
                      *[P(a) for a in outs], int(Niter) if Niter else 0, int(bool(l_use_skin)), P(rad[0]), P(rad[1]),
                      P(t_s), ni, nj, C.byref(rep))
    if rc:
        _raise(rc)
    res = {k: v.reshape(shape, order="F") for k, v in zip(OUT_NAMES[:5], outs)}
    if t_s is not None:
        res["T_s"] = t_s.reshape(shape, order="F")
    if jt == 1:
        res["init_report"] = dict(n_masked=rep.n_masked, hum_type=HUM_TYPES.get(rep.hum_type))
    return res


def device_count():
    """Usable gfx950 devices (ab_device_count)."""
    return int(_lib.load().ab_device_count())


def synth_fields_device(Ni, Nj, j0=0, nj_local=None, precision="f64", device="cuda", with_rad=True):
    """SURVEY §8d synthetic fields generated straight into HBM (torch tensors)."""
    import torch
    lib = _lib.load()
    if nj_local is None:
        nj_local = Nj - j0
    n = Ni * nj_local
    dt = torch.float64 if precision == "f64" else torch.float32
    names = IN_NAMES if with_rad else IN_NAMES[:6]
    f = {k: torch.empty(n, dtype=dt, device=device) for k in names}
    rc = lib.ab_synth_fields_device(*[f[k].data_ptr() for k in IN_NAMES[:6]],
                                    f["rad_sw"].data_ptr() if with_rad else None,
                                    f["rad_lw"].data_ptr() if with_rad else None, Ni, j0, nj_local,
                                    AB_F64 if precision == "f64" else AB_F32,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        _raise(rc)
    return f


CALIB = {"fma_f64": 0, "hbm_copy": 1}


def calibrate(what="fma_f64", device=0, stream=None):
    """ab_calibrate (include/aerobulk_amd.h): a fixed device workload that depends on the box alone -> (ms of the timed launch, rate);
    rate in fp64 TFLOP/s ("fma_f64") or GB/s read + written ("hbm_copy")."""
    lib = _lib.load()
    ms, rate = C.c_double(0.), C.c_double(0.)
    rc = lib.ab_calibrate(CALIB[what], int(device), C.c_void_p(stream or 0), C.byref(ms), C.byref(rate))
    if rc:
        _raise(rc)
    return ms.value, rate.value


ICE_ALGOS = {"nemo": 1, "an05": 2, "lu12": 3, "lg15": 4, "easy": 5}
ICE_OUT = ("Cd", "Ch", "Ce", "t_zu", "q_zu", "Ub", "CdN", "ChN", "CeN", "z0", "u_star", "L", "UN10")
ICE_OUT_LG15 = ICE_OUT + ("CdN_frm",)      # TURB_ICE_LG15_IO's form-drag output (mod_blk_ice_lg15_io.f90:71)


def turb_ice(calgo, zt, zu, Ts_i, theta_zt, qs_i, q_zt, U_zu, frice=None, nb_iter=5, optional=ICE_OUT[6:], precision="f64", cxn=None):
    """TURB_ICE_NEMO / AN05 / LU12 / LG15 / EASY (src/ice/mod_blk_ice_*.f90) on flat arrays: numpy (host) or torch (device).
    Returns the dict of the six mandatory outputs plus the requested OPTIONAL ones.  `cxn` = (CdN, ChN, CeN) for "easy";
    "lg15_io" is TURB_ICE_LG15_IO over ice (mod_blk_ice_lg15_io.f90:69): LG15 plus the optional "CdN_frm"."""
    lib = _lib.load()
    all_out = ICE_OUT
    if calgo == "lg15_io":
        calgo, all_out = "lg15", ICE_OUT_LG15
    if calgo not in ICE_ALGOS:
        raise AerobulkError(3, f"sea-ice algorithm {calgo} is unknown")
    dtype = np.float64 if precision == "f64" else np.float32
    dev = _is_torch(Ts_i)
    n = int(Ts_i.numel() if dev else np.asarray(Ts_i).size)
    f = _lib.IceFields()
    keep = []
    for k, a in (("Ts_i", Ts_i), ("theta_zt", theta_zt), ("qs_i", qs_i), ("q_zt", q_zt), ("U_zu", U_zu), ("frice", frice)):
        p, kp = _ptr(a, dtype, n)
        keep.append(kp)
        setattr(f, k, p)
    names = ICE_OUT[:6] + tuple(k for k in all_out[6:] if k in optional)
    if dev:
        import torch
        out = {k: torch.empty(n, dtype=Ts_i.dtype, device=Ts_i.device) for k in names}
        stream = torch.cuda.current_stream().cuda_stream
    else:
        out = {k: np.empty(n, dtype=dtype) for k in names}
        stream = 0
    for k, a in out.items():
        setattr(f, k, _ptr(a, dtype, n)[0])
    if calgo == "easy":
        if cxn is None:
            raise AerobulkError(10, "TURB_ICE_EASY needs cxn = (CdN, ChN, CeN)")
        rc = lib.ab_turb_ice_easy(float(zt), float(zu), int(nb_iter), float(cxn[0]), float(cxn[1]), float(cxn[2]), C.byref(f), n,
                                  AB_F64 if precision == "f64" else AB_F32, AB_MEM_DEVICE if dev else AB_MEM_HOST, C.c_void_p(stream or 0))
    else:
        rc = lib.ab_turb_ice(ICE_ALGOS[calgo], float(zt), float(zu), int(nb_iter), C.byref(f), n, AB_F64 if precision == "f64" else AB_F32,
                             AB_MEM_DEVICE if dev else AB_MEM_HOST, C.c_void_p(stream or 0))
    if rc:
        _raise(rc)
    if dev:
        import torch
        torch.cuda.synchronize()
    return out


def turb_neutral_10m(calgo, U_N10, nb_iter=5, precision="f64"):
    """TURB_NEUTRAL_10M (mod_blk_neutral_10m.f90:33): dict CdN10, ChN10, CeN10, z0 from the neutral 10 m wind (numpy or torch)."""
    lib = _lib.load()
    dtype = np.float64 if precision == "f64" else np.float32
    dev = _is_torch(U_N10)
    n = int(U_N10.numel() if dev else np.asarray(U_N10).size)
    pu, keep = _ptr(U_N10, dtype, n)
    if dev:
        import torch
        out = {k: torch.empty(n, dtype=U_N10.dtype, device=U_N10.device) for k in ("CdN10", "ChN10", "CeN10", "z0")}
        stream = torch.cuda.current_stream().cuda_stream
    else:
        out = {k: np.empty(n, dtype=dtype) for k in ("CdN10", "ChN10", "CeN10", "z0")}
        stream = 0
    rc = lib.ab_turb_neutral_10m(ALGOS.get(calgo, 0), int(nb_iter), pu, *[_ptr(a, dtype, n)[0] for a in out.values()], n,
                                 AB_F64 if precision == "f64" else AB_F32, AB_MEM_DEVICE if dev else AB_MEM_HOST, C.c_void_p(stream or 0))
    if rc:
        _raise(rc)
    if dev:
        import torch
        torch.cuda.synchronize()
    return out


def phymbl(fn, inputs, par0=0.0, flag=0, n_out=1, want=None, par1=0.0):
    """One public function of the reference's mod_phymbl (mod_phymbl.f90:33-139) on arrays: `ab_phymbl` of the C ABI.

    fn: enum ab_phymbl_fn (include/aerobulk_amd.h); inputs: the function's array arguments in the reference's order, None for an
    OPTIONAL one that is not passed (numpy: host arrays, staged; torch CUDA tensors: used in place); par0 / flag: its scalar REAL /
    LOGICAL-or-INTEGER argument (par1: zu of FIRST_GUESS_COARE, whose par0 is zt).  Returns (list of n_out arrays — None where `want` says not wanted —, info) with info =
    (first cell beyond 10 N/m^2 or -1, its stress) for BULK_FORMULA.  Raises AerobulkError except for AB_ERR_TAU, which is reported
    through info like BULK_FORMULA_VCTR's STOP message."""
    lib = _lib.load()
    first = next(x for x in inputs if x is not None)
    dev = _is_torch(first)
    n = int(first.numel() if dev else np.asarray(first).size)
    pin = (C.c_void_p * len(inputs))()
    keep = []
    for i, x in enumerate(inputs):
        a, k = _ptr(x, np.float64, n)
        pin[i] = a
        keep.append(k)
    want = [True] * n_out if want is None else list(want)
    if dev:
        import torch
        outs = [torch.empty(n, dtype=torch.float64, device=first.device) if w else None for w in want]
        stream = torch.cuda.current_stream(first.device).cuda_stream
    else:
        outs = [np.empty(n, dtype=np.float64) if w else None for w in want]
        stream = 0
    pout = (C.c_void_p * n_out)(*[_ptr(o, np.float64, n)[0] for o in outs])
    par = (C.c_double * 2)(float(par0), float(par1))
    info = (C.c_double * 2)(-1.0, 0.0)
    rc = lib.ab_phymbl(int(fn), n, pin, len(inputs), pout, n_out, par, int(flag), AB_MEM_DEVICE if dev else AB_MEM_HOST,
                       C.c_void_p(stream or 0), info)
    if rc and rc != AB_ERR_TAU:
        _raise(rc)
    return outs, (int(info[0]), float(info[1]))
