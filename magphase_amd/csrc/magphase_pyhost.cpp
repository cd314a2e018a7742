// _mpx_pyhost -- the marshalling layer between the Python batch API and the host-side C ABI (include/magphase_hip.h:
// mpx_host_plan_analysis_batch, mpx_host_plan_synthesis_batch).  A launch of a corpus job hands over 64-256 utterances as
// a list of tuples of numpy arrays; walking that list in Python (ctypes pointers, shapes, dtypes, one list append per
// field) cost as much as the device needs for the launch.  Here the list is walked once through the buffer protocol
// (no numpy headers, no copies), the pointers go straight to the C ABI, and the interpreter lock is released for the
// whole native call -- so a planner thread can prepare launch i + 1 while the main thread enqueues launch i
// (magphase_amd/engine.py: Engine.prepare_*; the reference's model is one worker per utterance with nothing shared,
// libutils.py:32-63).  Nothing here computes: every number comes from libmagphase_hip.so.
//
// A marshal function returns None when the batch is not in the plain shape it handles (other dtypes, non-contiguous
// arrays, Python lists): the caller then takes the generic Python path, which converts -- or raises what it raises.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/magphase_hip.h"

namespace {

struct Held {   // buffer views kept alive (and their exporters pinned) for the lifetime of a marshal object
    std::vector<Py_buffer> views;
    ~Held() {
        for (Py_buffer& v : views) PyBuffer_Release(&v);
    }
    // 1-D / 2-D C-contiguous view of obj with one of the formats in `fmts`; returns the index of the format or -1
    int take(PyObject* obj, const char* fmts, int ndim, Py_buffer** out) {
        Py_buffer v;
        if (PyObject_GetBuffer(obj, &v, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) != 0) {
            PyErr_Clear();
            return -1;
        }
        const char* f = v.format ? v.format : "B";
        if (*f == '<' || *f == '=' || *f == '@') ++f;
        const char* hit = (f[0] && !f[1]) ? strchr(fmts, f[0]) : nullptr;
        if (!hit || v.ndim != ndim) {
            PyBuffer_Release(&v);
            return -1;
        }
        views.push_back(v);
        *out = &views.back();
        return (int)(hit - fmts);
    }
};

struct AnaMarshal : Held {
    std::vector<const void*> pcm;
    std::vector<int32_t> kind;
    std::vector<int64_t> n_smpls, n_ep;
    std::vector<double> fs;
    std::vector<const double*> pm, voi;
    int64_t total_smpls = 0, total_ep = 0;
    bool all_i16 = true;
};

struct SynMarshal : Held {
    std::vector<const void*> mag, real, imag, lf0;
    std::vector<int32_t> kind, lf0_kind;
    std::vector<int64_t> n_rows;
    int64_t R = 0;
    int32_t mag_dim = 0, phase_dim = 0;
};

template <typename T>
void capsule_free(PyObject* c) {
    delete static_cast<T*>(PyCapsule_GetPointer(c, nullptr));
}

bool as_double(PyObject* o, double* out) {
    const double v = PyFloat_AsDouble(o);
    if (v == -1.0 && PyErr_Occurred()) {
        PyErr_Clear();
        return false;
    }
    *out = v;
    return true;
}

// item k of a tuple or list (borrowed), or null
PyObject* item(PyObject* seq, Py_ssize_t k) {
    if (PyTuple_Check(seq)) return k < PyTuple_GET_SIZE(seq) ? PyTuple_GET_ITEM(seq, k) : nullptr;
    if (PyList_Check(seq)) return k < PyList_GET_SIZE(seq) ? PyList_GET_ITEM(seq, k) : nullptr;
    return nullptr;
}
Py_ssize_t length(PyObject* seq) {
    if (PyTuple_Check(seq)) return PyTuple_GET_SIZE(seq);
    if (PyList_Check(seq)) return PyList_GET_SIZE(seq);
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// analysis: utts = [(v_sig int16 | float32 | float64 [n], fs, v_pm_sec float64 [E], v_voi float64 [E]), ...]
// ---------------------------------------------------------------------------------------------------------------------
PyObject* analysis_marshal(PyObject*, PyObject* utts) {
    const Py_ssize_t U = length(utts);
    if (U <= 0) Py_RETURN_NONE;
    AnaMarshal* m = new AnaMarshal;
    m->views.reserve((size_t)U * 3);
    for (Py_ssize_t u = 0; u < U; ++u) {
        PyObject* t = item(utts, u);
        if (!t || length(t) != 4) goto plain;
        {
            Py_buffer *b_sig, *b_pm, *b_voi;
            const int k = m->take(item(t, 0), "hfd", 1, &b_sig);
            double rate;
            if (k < 0 || !as_double(item(t, 1), &rate)) goto plain;
            if (m->take(item(t, 2), "d", 1, &b_pm) < 0 || m->take(item(t, 3), "d", 1, &b_voi) < 0) goto plain;
            if (b_pm->shape[0] != b_voi->shape[0]) goto plain;
            m->pcm.push_back(b_sig->buf);
            m->kind.push_back(k);
            m->n_smpls.push_back((int64_t)b_sig->shape[0]);
            m->fs.push_back(rate);
            m->pm.push_back((const double*)b_pm->buf);
            m->voi.push_back((const double*)b_voi->buf);
            m->n_ep.push_back((int64_t)b_pm->shape[0]);
            m->total_smpls += (int64_t)b_sig->shape[0];
            m->total_ep += (int64_t)b_pm->shape[0];
            m->all_i16 = m->all_i16 && k == 0;
        }
    }
    return PyCapsule_New(m, nullptr, capsule_free<AnaMarshal>);
plain:
    delete m;
    Py_RETURN_NONE;
}

PyObject* analysis_info(PyObject*, PyObject* cap) {
    AnaMarshal* m = static_cast<AnaMarshal*>(PyCapsule_GetPointer(cap, nullptr));
    if (!m) return nullptr;
    return Py_BuildValue("nLLO", (Py_ssize_t)m->pcm.size(), (long long)m->total_smpls, (long long)m->total_ep,
                         m->all_i16 ? Py_True : Py_False);
}

// writable buffer of at least `need` bytes, or null with an exception set (None -> null without one if optional)
struct OutBuf {
    Py_buffer v;
    bool held = false;
    ~OutBuf() {
        if (held) PyBuffer_Release(&v);
    }
    bool get(PyObject* o, int64_t need, bool optional, void** p) {
        *p = nullptr;
        if (o == Py_None) {
            if (optional) return true;
            PyErr_SetString(PyExc_TypeError, "output buffer is None");
            return false;
        }
        if (PyObject_GetBuffer(o, &v, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return false;
        held = true;
        if ((int64_t)v.len < need) {
            PyErr_SetString(PyExc_ValueError, "output buffer too small");
            return false;
        }
        *p = v.buf;
        return true;
    }
};

// analysis_run(cap, stage_addr, stage_kind, pos, left32, right32, voi32, pm, left64, f0, f0_med, frame_off, fft_len,
//              long_frame, long_len, n_threads) -> (n_frames or negative code, n_long)
PyObject* analysis_run(PyObject*, PyObject* args) {
    PyObject *cap, *o[11];
    unsigned long long stage_addr;
    int stage_kind, fft_len, n_threads;
    if (!PyArg_ParseTuple(args, "OKiOOOOOOOOOiOOi", &cap, &stage_addr, &stage_kind, &o[0], &o[1], &o[2], &o[3], &o[4], &o[5],
                          &o[6], &o[7], &o[8], &fft_len, &o[9], &o[10], &n_threads))
        return nullptr;
    AnaMarshal* m = static_cast<AnaMarshal*>(PyCapsule_GetPointer(cap, nullptr));
    if (!m) return nullptr;
    const int64_t E = m->total_ep, U = (int64_t)m->pcm.size();
    OutBuf b[11];
    void* p[11];
    const int64_t need[11] = {8 * E, 4 * E, 4 * E, 4 * E, 8 * E, 8 * E, 8 * E, 8 * E, 8 * (U + 1), 0, 0};
    const bool opt[11] = {false, false, false, true, false, false, false, true, false, true, true};
    for (int k = 0; k < 11; ++k)
        if (!b[k].get(o[k], need[k], opt[k], &p[k])) return nullptr;
    int64_t long_cap = 0;
    if (p[9] && p[10]) long_cap = (int64_t)((b[9].v.len < b[10].v.len ? b[9].v.len : b[10].v.len) / 8);
    int64_t n_long = 0, F;
    Py_BEGIN_ALLOW_THREADS
    F = mpx_host_plan_analysis_batch((int32_t)U, m->pcm.data(), m->kind.data(), m->n_smpls.data(), m->fs.data(),
                                     m->pm.data(), m->voi.data(), m->n_ep.data(), (void*)(uintptr_t)stage_addr, stage_kind,
                                     (int64_t*)p[0], (int32_t*)p[1], (int32_t*)p[2], (float*)p[3], (int64_t*)p[4],
                                     (int64_t*)p[5], (double*)p[6], (double*)p[7], (int64_t*)p[8], fft_len,
                                     (int64_t*)p[9], (int64_t*)p[10], long_cap, &n_long, n_threads);
    Py_END_ALLOW_THREADS
    return Py_BuildValue("LL", (long long)F, (long long)n_long);
}

// ---------------------------------------------------------------------------------------------------------------------
// synthesis: utts = [(m_mag_mel_log [rows x mag_dim], m_real_mel [rows x phase_dim], m_imag_mel, v_lf0 [rows]), ...]
// ---------------------------------------------------------------------------------------------------------------------
PyObject* synthesis_marshal(PyObject*, PyObject* utts) {
    const Py_ssize_t U = length(utts);
    if (U <= 0) Py_RETURN_NONE;
    SynMarshal* m = new SynMarshal;
    m->views.reserve((size_t)U * 4);
    for (Py_ssize_t u = 0; u < U; ++u) {
        PyObject* t = item(utts, u);
        if (!t || length(t) != 4) goto plain;
        {
            Py_buffer *bm, *br, *bi, *bl;
            const int km = m->take(item(t, 0), "fd", 2, &bm);
            if (km < 0) goto plain;
            const int kr = m->take(item(t, 1), "fd", 2, &br), ki = m->take(item(t, 2), "fd", 2, &bi);
            const int kl = m->take(item(t, 3), "fd", 1, &bl);
            if (kr != km || ki != km || kl < 0) goto plain;
            const int64_t rows = (int64_t)bm->shape[0];
            // (mismatching frame counts / dimensions: the generic path raises the reference's ValueError)
            if (br->shape[0] != rows || bi->shape[0] != rows || bl->shape[0] != rows || br->shape[1] != bi->shape[1]) goto plain;
            if (u == 0) {
                m->mag_dim = (int32_t)bm->shape[1];
                m->phase_dim = (int32_t)br->shape[1];
            } else if (bm->shape[1] != m->mag_dim || br->shape[1] != m->phase_dim) {
                goto plain;
            }
            m->mag.push_back(bm->buf), m->real.push_back(br->buf), m->imag.push_back(bi->buf), m->lf0.push_back(bl->buf);
            m->kind.push_back(km + 1);
            m->lf0_kind.push_back(kl);
            m->n_rows.push_back(rows);
            m->R += rows;
        }
    }
    if (m->mag_dim < 1 || m->phase_dim < 1) goto plain;
    return PyCapsule_New(m, nullptr, capsule_free<SynMarshal>);
plain:
    delete m;
    Py_RETURN_NONE;
}

PyObject* synthesis_info(PyObject*, PyObject* cap) {
    SynMarshal* m = static_cast<SynMarshal*>(PyCapsule_GetPointer(cap, nullptr));
    if (!m) return nullptr;
    return Py_BuildValue("nLii", (Py_ssize_t)m->mag.size(), (long long)m->R, (int)m->mag_dim, (int)m->phase_dim);
}

// synthesis_lf0(cap, out float64 [R]): the utterances' lf0 vectors end to end (float32 widened: exact)
PyObject* synthesis_lf0(PyObject*, PyObject* args) {
    PyObject *cap, *out;
    if (!PyArg_ParseTuple(args, "OO", &cap, &out)) return nullptr;
    SynMarshal* m = static_cast<SynMarshal*>(PyCapsule_GetPointer(cap, nullptr));
    if (!m) return nullptr;
    OutBuf b;
    void* p;
    if (!b.get(out, 8 * m->R, false, &p)) return nullptr;
    double* d = (double*)p;
    for (size_t u = 0; u < m->lf0.size(); ++u) {
        const int64_t n = m->n_rows[u];
        if (m->lf0_kind[u] == 1) {
            memcpy(d, m->lf0[u], (size_t)n * 8);
        } else {
            const float* s = (const float*)m->lf0[u];
            for (int64_t i = 0; i < n; ++i) d[i] = (double)s[i];
        }
        d += n;
    }
    Py_RETURN_NONE;
}

// synthesis_run(cap, stage_addr, f0, fs, fft_len, b_const_rate, b_voi_ap_win, n_slots, wcum | None, wsum, want_tiles, desc,
//               desc_off, v_shift, v_pm, voiced_host, frame_off, ns_len, out_start, out_len, runs_host, counts, n_threads)
PyObject* synthesis_run(PyObject*, PyObject* args) {
    PyObject *cap, *o_f0, *o_wcum, *o[11];
    unsigned long long stage_addr;
    double fs, wsum;
    int fft_len, b_const, b_voi, n_slots, want_tiles, n_threads;
    if (!PyArg_ParseTuple(args, "OKOdiiiiOdiOOOOOOOOOOOi", &cap, &stage_addr, &o_f0, &fs, &fft_len, &b_const, &b_voi, &n_slots,
                          &o_wcum, &wsum, &want_tiles, &o[0], &o[1], &o[2], &o[3], &o[4], &o[5], &o[6], &o[7], &o[8], &o[9],
                          &o[10], &n_threads))
        return nullptr;
    SynMarshal* m = static_cast<SynMarshal*>(PyCapsule_GetPointer(cap, nullptr));
    if (!m) return nullptr;
    const int64_t U = (int64_t)m->mag.size(), cap_f = 2 * m->R + 2 * U;
    Py_buffer vf0, vw;
    if (PyObject_GetBuffer(o_f0, &vf0, PyBUF_C_CONTIGUOUS) != 0) return nullptr;
    if ((int64_t)vf0.len < 8 * m->R) {
        PyBuffer_Release(&vf0);
        PyErr_SetString(PyExc_ValueError, "f0 too short");
        return nullptr;
    }
    const double* wcum = nullptr;
    bool have_w = false;
    if (o_wcum != Py_None) {
        if (PyObject_GetBuffer(o_wcum, &vw, PyBUF_C_CONTIGUOUS) != 0) {
            PyBuffer_Release(&vf0);
            return nullptr;
        }
        have_w = true;
        if ((int64_t)vw.len < 8 * ((int64_t)n_slots + 1)) {
            PyBuffer_Release(&vf0), PyBuffer_Release(&vw);
            PyErr_SetString(PyExc_ValueError, "wcum too short");
            return nullptr;
        }
        wcum = (const double*)vw.buf;
    }
    OutBuf b[11];
    void* p[11];
    const int64_t need[11] = {0, 8 * 18, 8 * cap_f, 8 * cap_f, 4 * cap_f, 8 * (U + 1), 8 * U, 8 * U, 8 * U,
                              (int64_t)sizeof(mpx_ola_run) * (U + n_slots + 1), 8 * 8};
    bool ok = true;
    for (int k = 0; k < 11 && ok; ++k) ok = b[k].get(o[k], need[k], false, &p[k]);
    int64_t F = -1;
    if (ok) {
        const int64_t desc_cap = (int64_t)b[0].v.len;
        const int64_t runs_cap = (int64_t)(b[9].v.len / (Py_ssize_t)sizeof(mpx_ola_run));
        Py_BEGIN_ALLOW_THREADS
        F = mpx_host_plan_synthesis_batch((int32_t)U, m->mag.data(), m->real.data(), m->imag.data(), m->kind.data(),
                                          m->n_rows.data(), m->mag_dim, m->phase_dim, (float*)(uintptr_t)stage_addr,
                                          (const double*)vf0.buf, fs, fft_len, b_const, b_voi, n_slots, wcum, wsum, want_tiles,
                                          (uint8_t*)p[0], desc_cap, (int64_t*)p[1], (int64_t*)p[2], (int64_t*)p[3],
                                          (int32_t*)p[4], (int64_t*)p[5], (int64_t*)p[6], (int64_t*)p[7], (int64_t*)p[8],
                                          (mpx_ola_run*)p[9], runs_cap, (int64_t*)p[10], n_threads);
        Py_END_ALLOW_THREADS
    }
    PyBuffer_Release(&vf0);
    if (have_w) PyBuffer_Release(&vw);
    if (!ok) return nullptr;
    return PyLong_FromLongLong((long long)F);
}

PyMethodDef kMethods[] = {
    {"analysis_marshal", analysis_marshal, METH_O, "list of (v_sig, fs, v_pm_sec, v_voi) -> marshal object or None"},
    {"analysis_info", analysis_info, METH_O, "(n_utts, total samples, total epochs, all int16)"},
    {"analysis_run", analysis_run, METH_VARARGS, "mpx_host_plan_analysis_batch on a marshal object"},
    {"synthesis_marshal", synthesis_marshal, METH_O, "list of (mag, real, imag, lf0) -> marshal object or None"},
    {"synthesis_info", synthesis_info, METH_O, "(n_utts, total rows, mag_dim, phase_dim)"},
    {"synthesis_lf0", synthesis_lf0, METH_VARARGS, "the utterances' lf0 vectors end to end, float64"},
    {"synthesis_run", synthesis_run, METH_VARARGS, "mpx_host_plan_synthesis_batch on a marshal object"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef kModule = {PyModuleDef_HEAD_INIT, "_mpx_pyhost", "marshalling layer of magphase_amd's batch API", -1, kMethods,
                       nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__mpx_pyhost(void) { return PyModule_Create(&kModule); }
