/* rlpyt_amd._envloop -- the per-environment bookkeeping of a sampler worker's time step in C.
 *
 * What it replaces: the body of EnvRunner.step_all (rlpyt_amd/samplers/gpu.py; the worker side of
 * rlpyt/samplers/parallel/gpu/collectors.py:18-50 -- step every env with the action the master
 * published, write observation / reward / done / env_info of the step into the shared step buffer)
 * together with the stock trajectory statistics (rlpyt/samplers/collections.py:30-56 and
 * rlpyt/envs/atari/atari_env.py:24-30).  Once the device side of a time step is ~100 us the rollout is
 * priced in host CPU-seconds per env step under the box's CPU quota, and ~1/3 of those were
 * interpreter overhead of that loop body (dict updates of the TrajInfo, half a dozen numpy scalar
 * stores, attribute probing, tuple unpacking).  env.step() itself stays a Python call.
 *
 * Only the mid-batch-reset collector with the stock TrajInfo / AtariTrajInfo takes this path; everything
 * else keeps the Python loop.  The arithmetic of the statistics follows numpy's promotion rules for
 * the reference's per-step updates: with np.float32 rewards, Return and DiscountedReturn accumulate in
 * float32 (the discount factor in float64, cast per step); the first reward of another type switches
 * the accumulators to float64 for good.
 *
 * This is host logic (no device code); built by csrc/Makefile into rlpyt_amd/_envloop.so.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <limits.h>
#include <linux/futex.h>
#include <stdint.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

/* The step hand-off words of csrc/hostsync.cpp (rlpyt_seq_wait / rlpyt_seq_arrive), restated here so
 * that a worker's wait -> step -> arrive of one pipeline group is ONE call from Python. */
static inline uint64_t mono_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
static inline int seq_reached(uint32_t cur, uint32_t target) { return (int32_t)(cur - target) >= 0; }
/* Returns 0 once the word reached `target`, 1 after a futex wait ended without that (timeout of
 * 50 ms, or EINTR from a signal): the caller -- which holds no GIL while in here -- then lets the
 * interpreter run its signal handlers before waiting again. */
static int seq_wait(uint32_t* word, uint32_t target, int spin_iters) {
  for (int i = 0; i < spin_iters; ++i) {
    if (seq_reached(__atomic_load_n(word, __ATOMIC_ACQUIRE), target)) return 0;
    __builtin_ia32_pause();
  }
  const uint32_t cur = __atomic_load_n(word, __ATOMIC_ACQUIRE);
  if (seq_reached(cur, target)) return 0;
  struct timespec ts = {0, 50 * 1000 * 1000};   /* re-check at least every 50 ms */
  syscall(SYS_futex, word, FUTEX_WAIT, cur, &ts, NULL, 0);
  return seq_reached(__atomic_load_n(word, __ATOMIC_ACQUIRE), target) ? 0 : 1;
}
static void seq_arrive(uint32_t* word, uint32_t wake_at) {
  const uint32_t now = __atomic_add_fetch(word, 1u, __ATOMIC_ACQ_REL);
  if (seq_reached(now, wake_at)) syscall(SYS_futex, word, FUTEX_WAKE, INT_MAX, NULL, NULL, 0);
}

typedef struct {
  Py_buffer view;
  int ok;
} Buf;

static int buf_get(PyObject* o, Buf* b, int writable) {
  b->ok = 0;
  if (o == Py_None) return 0;
  if (PyObject_GetBuffer(o, &b->view, (writable ? PyBUF_WRITABLE : 0) | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0)
    return -1;
  b->ok = 1;
  return 0;
}
static void buf_rel(Buf* b) {
  if (b->ok) PyBuffer_Release(&b->view);
  b->ok = 0;
}

#define MAX_INFO 8
enum { K_F32 = 1, K_F64, K_BOOL, K_I32, K_I64, K_U8 };

typedef struct {
  PyObject_HEAD
  PyObject* envs;        /* list */
  PyObject* on_done;     /* callable(b, final_obs) -> reset observation */
  PyObject* f32_type;    /* numpy.float32 */
  Py_ssize_t n;
  Buf act, rew, done, frame, reset, obs;
  Py_ssize_t frame_bytes, obs_bytes;
  int n_info;
  Buf info[MAX_INFO];          /* [T, n] arrays, possibly column slices of the sampler's [T, B] */
  int info_kind[MAX_INFO];
  Py_ssize_t info_T, info_s0[MAX_INFO], info_s1[MAX_INFO];
  /* trajectory statistics, one entry per env */
  Buf len_, nz, g, score;      /* i64, i64, f64, f64 (score optional) */
  Buf ret32, disc32, ret64, disc64;
  int f64_mode;                /* 0: float32 accumulators are live, 1: float64 */
  int wait_reset;              /* 1: the wait-reset collector (collectors.py:73-126): a finished env
                                * idles until the batch ends, its observation row is blanked */
  Buf force_full;              /* wait-reset only: bool [n], "the next observation starts a new stack" */
  double discount;
  int has_td, has_score;       /* -1 unknown, 0 no, 1 yes -- for infos of type `info_type` */
  PyTypeObject* info_type;     /* borrowed: only compared, never dereferenced */
  PyObject *s_step, *s_traj_done, *s_game_score;
  PyObject** small_ints;       /* cached action objects 0..63 */
  /* where a worker's time goes (step_synced only): waiting for the master's actions vs stepping */
  uint64_t wait_ns, step_ns, n_synced;
  uint64_t wake_ns, n_waited;  /* post -> running again, for the waits that really waited */
} EnvLoop;

static int kind_of(const char* fmt, Py_ssize_t itemsize) {
  if (!fmt) return 0;
  while (*fmt == '<' || *fmt == '=' || *fmt == '@' || *fmt == '|') ++fmt;
  switch (*fmt) {
    case 'f': return itemsize == 4 ? K_F32 : 0;
    case 'd': return K_F64;
    case '?': return K_BOOL;
    case 'i': return itemsize == 4 ? K_I32 : 0;
    case 'l': case 'q': return itemsize == 8 ? K_I64 : 0;
    case 'B': return K_U8;
    default: return 0;
  }
}

static void EnvLoop_dealloc(EnvLoop* self) {
  Py_XDECREF(self->envs);
  Py_XDECREF(self->on_done);
  buf_rel(&self->force_full);
  Py_XDECREF(self->f32_type);
  Py_XDECREF(self->s_step);
  Py_XDECREF(self->s_traj_done);
  Py_XDECREF(self->s_game_score);
  buf_rel(&self->act); buf_rel(&self->rew); buf_rel(&self->done); buf_rel(&self->frame);
  buf_rel(&self->reset); buf_rel(&self->obs);
  for (int i = 0; i < MAX_INFO; ++i) buf_rel(&self->info[i]);
  buf_rel(&self->len_); buf_rel(&self->nz); buf_rel(&self->g); buf_rel(&self->score);
  buf_rel(&self->ret32); buf_rel(&self->disc32); buf_rel(&self->ret64); buf_rel(&self->disc64);
  if (self->small_ints) {
    for (int i = 0; i < 64; ++i) Py_XDECREF(self->small_ints[i]);
    PyMem_Free(self->small_ints);
  }
  Py_TYPE(self)->tp_free((PyObject*)self);
}

static int expect(Buf* b, int kind, Py_ssize_t count, const char* name) {
  if (!b->ok) { PyErr_Format(PyExc_TypeError, "EnvLoop: %s is required", name); return -1; }
  if (kind_of(b->view.format, b->view.itemsize) != kind || b->view.len != count * b->view.itemsize) {
    PyErr_Format(PyExc_TypeError, "EnvLoop: %s has the wrong dtype or size (format %s, %zd bytes)",
                 name, b->view.format ? b->view.format : "?", b->view.len);
    return -1;
  }
  return 0;
}

static int EnvLoop_init(EnvLoop* self, PyObject* args, PyObject* kw) {
  static char* kwl[] = {"envs", "action", "reward", "done", "frame", "reset", "observation",
                        "info_arrays", "length", "ret32", "nonzero", "disc32", "ret64", "disc64",
                        "cur_discount", "score", "discount", "f64_mode", "on_done", "float32_type",
                        "wait_reset", "force_full", NULL};
  PyObject *envs, *act, *rew, *done, *frame, *reset, *obs, *infos, *len_, *ret32, *nz, *disc32, *ret64,
      *disc64, *g, *score, *on_done, *f32t, *force_full = Py_None;
  double discount;
  int f64_mode, wait_reset = 0;
  if (!PyArg_ParseTupleAndKeywords(args, kw, "OOOOOOOOOOOOOOOOdiOO|iO", kwl, &envs, &act, &rew, &done,
                                   &frame, &reset, &obs, &infos, &len_, &ret32, &nz, &disc32, &ret64,
                                   &disc64, &g, &score, &discount, &f64_mode, &on_done, &f32t,
                                   &wait_reset, &force_full))
    return -1;
  if (!PyList_Check(envs)) { PyErr_SetString(PyExc_TypeError, "EnvLoop: envs must be a list"); return -1; }
  self->n = PyList_GET_SIZE(envs);
  Py_INCREF(envs); self->envs = envs;
  Py_INCREF(on_done); self->on_done = on_done;
  Py_INCREF(f32t); self->f32_type = f32t;
  self->discount = discount;
  self->f64_mode = f64_mode;
  self->wait_reset = wait_reset;
  self->force_full.ok = 0;
  if (buf_get(force_full, &self->force_full, 1)) return -1;
  self->has_td = self->has_score = -1;
  self->info_type = NULL;
  self->wait_ns = self->step_ns = self->n_synced = self->wake_ns = self->n_waited = 0;
  if (buf_get(act, &self->act, 0) || buf_get(rew, &self->rew, 1) || buf_get(done, &self->done, 1) ||
      buf_get(frame, &self->frame, 1) || buf_get(reset, &self->reset, 1) || buf_get(obs, &self->obs, 1) ||
      buf_get(len_, &self->len_, 1) || buf_get(nz, &self->nz, 1) || buf_get(g, &self->g, 1) ||
      buf_get(score, &self->score, 1) || buf_get(ret32, &self->ret32, 1) ||
      buf_get(disc32, &self->disc32, 1) || buf_get(ret64, &self->ret64, 1) ||
      buf_get(disc64, &self->disc64, 1))
    return -1;
  const Py_ssize_t n = self->n;
  if (expect(&self->act, K_I64, n, "action") || expect(&self->rew, K_F32, n, "reward") ||
      expect(&self->done, K_BOOL, n, "done") || expect(&self->len_, K_I64, n, "length") ||
      expect(&self->nz, K_I64, n, "nonzero") || expect(&self->g, K_F64, n, "cur_discount") ||
      expect(&self->ret32, K_F32, n, "ret32") || expect(&self->disc32, K_F32, n, "disc32") ||
      expect(&self->ret64, K_F64, n, "ret64") || expect(&self->disc64, K_F64, n, "disc64"))
    return -1;
  if (self->score.ok && expect(&self->score, K_F64, n, "score")) return -1;
  if (self->force_full.ok && expect(&self->force_full, K_BOOL, n, "force_full")) return -1;
  if (!self->obs.ok || self->obs.view.len % n) {
    PyErr_SetString(PyExc_TypeError, "EnvLoop: observation buffer required, [n, ...]");
    return -1;
  }
  self->obs_bytes = self->obs.view.len / n;
  self->frame_bytes = 0;
  if (self->frame.ok) {
    if (!self->reset.ok || expect(&self->reset, K_BOOL, n, "reset")) return -1;
    if (self->frame.view.len % n) { PyErr_SetString(PyExc_TypeError, "EnvLoop: frame buffer [n, ...]"); return -1; }
    self->frame_bytes = self->frame.view.len / n;
    if (self->frame_bytes > self->obs_bytes || self->obs_bytes % self->frame_bytes) {
      PyErr_SetString(PyExc_TypeError, "EnvLoop: a frame must be a whole slice of an observation");
      return -1;
    }
  }
  self->n_info = 0;
  self->info_T = 0;
  if (infos != Py_None) {
    PyObject* seq = PySequence_Fast(infos, "EnvLoop: info_arrays must be a sequence");
    if (!seq) return -1;
    const Py_ssize_t k = PySequence_Fast_GET_SIZE(seq);
    if (k > MAX_INFO) { Py_DECREF(seq); PyErr_SetString(PyExc_TypeError, "EnvLoop: too many env_info fields"); return -1; }
    for (Py_ssize_t i = 0; i < k; ++i) {
      Buf* bi = &self->info[i];
      bi->ok = 0;
      if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(seq, i), &bi->view,
                             PyBUF_WRITABLE | PyBUF_STRIDES | PyBUF_FORMAT) != 0) {
        Py_DECREF(seq);
        return -1;
      }
      bi->ok = 1;
      const int kind = kind_of(bi->view.format, bi->view.itemsize);
      if (!kind || bi->view.ndim != 2 || bi->view.shape[1] != n) {
        Py_DECREF(seq);
        PyErr_SetString(PyExc_TypeError, "EnvLoop: env_info arrays must be [T, n] of a plain dtype");
        return -1;
      }
      self->info_kind[i] = kind;
      self->info_T = bi->view.shape[0];
      self->info_s0[i] = bi->view.strides[0];
      self->info_s1[i] = bi->view.strides[1];
    }
    self->n_info = (int)k;
    Py_DECREF(seq);
  }
  self->s_step = PyUnicode_InternFromString("step");
  self->s_traj_done = PyUnicode_InternFromString("traj_done");
  self->s_game_score = PyUnicode_InternFromString("game_score");
  self->small_ints = (PyObject**)PyMem_Calloc(64, sizeof(PyObject*));
  if (!self->s_step || !self->s_traj_done || !self->s_game_score || !self->small_ints) return -1;
  for (int i = 0; i < 64; ++i)
    if (!(self->small_ints[i] = PyLong_FromLong(i))) return -1;
  return 0;
}

static void store_info(EnvLoop* self, int k, Py_ssize_t t, Py_ssize_t b, PyObject* v, int* err) {
  char* base = (char*)self->info[k].view.buf + t * self->info_s0[k] + b * self->info_s1[k];
  const Py_ssize_t idx = 0;
  switch (self->info_kind[k]) {
    case K_F32: { double x = PyFloat_AsDouble(v); if (x == -1.0 && PyErr_Occurred()) { *err = 1; return; } ((float*)base)[idx] = (float)x; break; }
    case K_F64: { double x = PyFloat_AsDouble(v); if (x == -1.0 && PyErr_Occurred()) { *err = 1; return; } ((double*)base)[idx] = x; break; }
    case K_BOOL: { int x = PyObject_IsTrue(v); if (x < 0) { *err = 1; return; } ((uint8_t*)base)[idx] = (uint8_t)x; break; }
    case K_U8: { long x = PyLong_AsLong(v); if (x == -1 && PyErr_Occurred()) { *err = 1; return; } ((uint8_t*)base)[idx] = (uint8_t)x; break; }
    case K_I32: { long x = PyLong_AsLong(v); if (x == -1 && PyErr_Occurred()) { *err = 1; return; } ((int32_t*)base)[idx] = (int32_t)x; break; }
    case K_I64: { long long x = PyLong_AsLongLong(v); if (x == -1 && PyErr_Occurred()) { *err = 1; return; } ((int64_t*)base)[idx] = (int64_t)x; break; }
  }
}

/* copy `nbytes` from the END of observation object `o` (contiguous buffer) -- the newest frame of a
 * frame stack -- or all of it */
static int copy_obs(PyObject* o, char* dst_full, Py_ssize_t obs_bytes, char* dst_frame,
                    Py_ssize_t frame_bytes, int want_full) {
  Py_buffer v;
  if (PyObject_GetBuffer(o, &v, PyBUF_C_CONTIGUOUS) != 0) return -1;
  if (v.len != obs_bytes) {
    PyBuffer_Release(&v);
    PyErr_Format(PyExc_ValueError, "EnvLoop: observation of %zd bytes, step buffer row has %zd", v.len,
                 obs_bytes);
    return -1;
  }
  if (dst_frame) memcpy(dst_frame, (const char*)v.buf + obs_bytes - frame_bytes, (size_t)frame_bytes);
  if (want_full) memcpy(dst_full, v.buf, (size_t)obs_bytes);
  PyBuffer_Release(&v);
  return 0;
}

/* step(t, lazy) -> None.  Steps every env once; see the header. */
static PyObject* EnvLoop_step(EnvLoop* self, PyObject* args) {
  Py_ssize_t t;
  int lazy;
  if (!PyArg_ParseTuple(args, "np", &t, &lazy)) return NULL;
  if (self->n_info && (t < 0 || t >= self->info_T)) {
    PyErr_SetString(PyExc_IndexError, "EnvLoop.step: t outside the env_info arrays");
    return NULL;
  }
  const int64_t* act = (const int64_t*)self->act.view.buf;
  float* rew = (float*)self->rew.view.buf;
  uint8_t* done = (uint8_t*)self->done.view.buf;
  uint8_t* reset = self->reset.ok ? (uint8_t*)self->reset.view.buf : NULL;
  char* frame = self->frame.ok ? (char*)self->frame.view.buf : NULL;
  char* obs = (char*)self->obs.view.buf;
  int64_t* len_ = (int64_t*)self->len_.view.buf;
  int64_t* nz = (int64_t*)self->nz.view.buf;
  double* g = (double*)self->g.view.buf;
  double* score = self->score.ok ? (double*)self->score.view.buf : NULL;
  float *ret32 = (float*)self->ret32.view.buf, *disc32 = (float*)self->disc32.view.buf;
  double *ret64 = (double*)self->ret64.view.buf, *disc64 = (double*)self->disc64.view.buf;
  if (!frame) lazy = 0;

  uint8_t* force_full = self->force_full.ok ? (uint8_t*)self->force_full.view.buf : NULL;
  for (Py_ssize_t b = 0; b < self->n; ++b) {
    if (self->wait_reset && done[b]) {
      /* wait-reset: a finished env idles with done = True and a blank reward until the batch ends
       * (collectors.py:85-91); nothing else of its rows is touched */
      rew[b] = 0.f;
      /* its blank stack was uploaded whole with the step that finished it; from here on "previous
       * stack shifted + blank newest frame" IS the blank stack: no upload per idle step (with the flag
       * left standing the master copied 33 KB per idle env and step -- a separate transfer each: the
       * R2D1 rollout's issue thread spent 48 us per group-step on them, profiles/r5_replay_sampler_breakdown.txt) */
      if (reset) reset[b] = 0;
      continue;
    }
    PyObject* env = PyList_GET_ITEM(self->envs, b);
    const int64_t a = act[b];
    PyObject* a_obj = (a >= 0 && a < 64) ? self->small_ints[a] : NULL;
    PyObject* a_new = NULL;
    if (!a_obj) { a_new = PyLong_FromLongLong(a); if (!a_new) return NULL; a_obj = a_new; }
    PyObject* res = PyObject_CallMethodObjArgs(env, self->s_step, a_obj, NULL);
    Py_XDECREF(a_new);
    if (!res) return NULL;
    if (!PyTuple_Check(res) || PyTuple_GET_SIZE(res) != 4) {
      Py_DECREF(res);
      PyErr_SetString(PyExc_TypeError, "EnvLoop: env.step must return (observation, reward, done, info)");
      return NULL;
    }
    PyObject* o = PyTuple_GET_ITEM(res, 0);
    PyObject* r_obj = PyTuple_GET_ITEM(res, 1);
    PyObject* d_obj = PyTuple_GET_ITEM(res, 2);
    PyObject* info = PyTuple_GET_ITEM(res, 3);
    const double r = PyFloat_AsDouble(r_obj);
    if (r == -1.0 && PyErr_Occurred()) { Py_DECREF(res); return NULL; }
    const int d = PyObject_IsTrue(d_obj);
    if (d < 0) { Py_DECREF(res); return NULL; }
    /* ---- trajectory statistics (collections.py:38-46) ---- */
    if (!self->f64_mode && (PyObject*)Py_TYPE(r_obj) != self->f32_type) {
      /* a reward that is not np.float32: numpy promotes the running sums to float64 from here on */
      for (Py_ssize_t i = 0; i < self->n; ++i) { ret64[i] = (double)ret32[i]; disc64[i] = (double)disc32[i]; }
      self->f64_mode = 1;
    }
    len_[b] += 1;
    if (self->f64_mode) {
      ret64[b] += r;
      disc64[b] += g[b] * r;
    } else {
      const float r32 = (float)r;
      ret32[b] += r32;
      const float prod = (float)g[b] * r32;
      disc32[b] += prod;
    }
    nz[b] += (r != 0.0);
    g[b] *= self->discount;
    /* ---- env_info ---- */
    int td = d;
    const int info_is_tuple = PyTuple_Check(info);
    const Py_ssize_t n_fields = info_is_tuple ? PyTuple_GET_SIZE(info) : 0;
    if (Py_TYPE(info) != self->info_type) {     /* another info class: probe its attributes afresh */
      self->info_type = Py_TYPE(info);
      self->has_td = self->has_score = -1;
    }
    if (n_fields > 0 || (info != Py_None && !info_is_tuple)) {
      if (self->has_td != 0) {
        PyObject* v = PyObject_GetAttr(info, self->s_traj_done);
        if (v) { td = PyObject_IsTrue(v); Py_DECREF(v); self->has_td = 1; if (td < 0) { Py_DECREF(res); return NULL; } }
        else { PyErr_Clear(); self->has_td = 0; }
      }
      if (score && self->has_score != 0) {
        PyObject* v = PyObject_GetAttr(info, self->s_game_score);
        if (v) {
          const double x = PyFloat_AsDouble(v);
          Py_DECREF(v);
          if (x == -1.0 && PyErr_Occurred()) { Py_DECREF(res); return NULL; }
          score[b] += x;
          self->has_score = 1;
        } else { PyErr_Clear(); self->has_score = 0; }
      }
      if (self->n_info) {
        int err = 0;
        const Py_ssize_t k = n_fields < self->n_info ? n_fields : self->n_info;
        for (Py_ssize_t i = 0; i < k && !err; ++i) store_info(self, (int)i, t, b, PyTuple_GET_ITEM(info, i), &err);
        if (err) { Py_DECREF(res); return NULL; }
      }
    }
    /* ---- end of a trajectory: record + reset through the Python callback ---- */
    int fresh = 0, blank = 0;
    PyObject* o_keep = NULL;
    if (!self->wait_reset) {
      if (td) {
        PyObject* b_obj = PyLong_FromSsize_t(b);
        if (!b_obj) { Py_DECREF(res); return NULL; }
        o_keep = PyObject_CallFunctionObjArgs(self->on_done, b_obj, o, NULL);
        Py_DECREF(b_obj);
        if (!o_keep) { Py_DECREF(res); return NULL; }
        o = o_keep;
        fresh = 1;
      }
    } else if (td || d) {
      /* wait-reset: on_done(b, final_obs, traj_done, done) records the trajectory / marks the env for
       * the reset between batches / holds the final observation; the env itself is NOT reset here and
       * a done env's observation row goes blank (collectors.py:96-103) */
      PyObject* b_obj = PyLong_FromSsize_t(b);
      if (!b_obj) { Py_DECREF(res); return NULL; }
      PyObject* r_ = PyObject_CallFunctionObjArgs(self->on_done, b_obj, o, td ? Py_True : Py_False,
                                                  d ? Py_True : Py_False, NULL);
      Py_DECREF(b_obj);
      if (!r_) { Py_DECREF(res); return NULL; }
      Py_DECREF(r_);
      if (d) { blank = 1; fresh = 1; }
    }
    if (force_full && force_full[b]) { fresh = 1; force_full[b] = 0; }
    /* ---- step buffer rows ---- */
    if (blank) {
      if (frame) memset(frame + b * self->frame_bytes, 0, (size_t)self->frame_bytes);
      memset(obs + b * self->obs_bytes, 0, (size_t)self->obs_bytes);
    } else if (copy_obs(o, obs + b * self->obs_bytes, self->obs_bytes, frame ? frame + b * self->frame_bytes : NULL,
                        self->frame_bytes, fresh || !lazy) != 0) {
      Py_XDECREF(o_keep);
      Py_DECREF(res);
      return NULL;
    }
    if (reset) reset[b] = (uint8_t)fresh;
    rew[b] = (float)r;
    done[b] = (uint8_t)d;
    Py_XDECREF(o_keep);
    Py_DECREF(res);
  }
  Py_RETURN_NONE;
}

/* step_synced(t, lazy, act_word, act_target, spin, obs_word, wake_at): wait until the master has
 * published action set `act_target`, step, report the arrival (the last arriver wakes the master). */
static PyObject* EnvLoop_step_synced(EnvLoop* self, PyObject* args) {
  Py_ssize_t t;
  int lazy, spin;
  unsigned long long act_word, obs_word;
  unsigned long act_target, wake_at;
  if (!PyArg_ParseTuple(args, "npKkiKk", &t, &lazy, &act_word, &act_target, &spin, &obs_word, &wake_at))
    return NULL;
  /* the wait releases the GIL (as the ctypes call it replaces did) and lets Python-level signal
   * handlers (SIGINT, a SIGTERM handler) run between futex sleeps */
  const uint64_t t0 = mono_ns();
  const int had_to_wait = !seq_reached(__atomic_load_n((uint32_t*)(uintptr_t)act_word, __ATOMIC_ACQUIRE),
                                       (uint32_t)act_target);
  for (;;) {
    int again;
    Py_BEGIN_ALLOW_THREADS
    again = seq_wait((uint32_t*)(uintptr_t)act_word, (uint32_t)act_target, spin);
    Py_END_ALLOW_THREADS
    if (!again) break;
    if (PyErr_CheckSignals() != 0) return NULL;
    spin = 0;
  }
  const uint64_t t1 = mono_ns();
  if (had_to_wait) {
    /* the master stamps CLOCK_MONOTONIC ns of its post next to the sequence word (csrc/serve.cpp) */
    const uint64_t posted = *(volatile uint64_t*)((char*)(uintptr_t)act_word + 8);
    if (posted != 0 && t1 >= posted && t1 - posted < 1000000000ull) {
      self->wake_ns += t1 - posted;
      self->n_waited += 1;
    }
  }
  PyObject* a2 = Py_BuildValue("(ni)", t, lazy);
  if (!a2) return NULL;
  PyObject* r = EnvLoop_step(self, a2);
  Py_DECREF(a2);
  if (!r) return NULL;
  Py_DECREF(r);
  seq_arrive((uint32_t*)(uintptr_t)obs_word, (uint32_t)wake_at);
  self->wait_ns += t1 - t0;
  self->step_ns += mono_ns() - t1;
  self->n_synced += 1;
  Py_RETURN_NONE;
}

/* timing() -> (wait_ns, step_ns, calls) accumulated by step_synced since the last call; resets. */
static PyObject* EnvLoop_timing(EnvLoop* self, PyObject* Py_UNUSED(ignored)) {
  PyObject* r = Py_BuildValue("(KKKKK)", (unsigned long long)self->wait_ns,
                              (unsigned long long)self->step_ns, (unsigned long long)self->n_synced,
                              (unsigned long long)self->wake_ns, (unsigned long long)self->n_waited);
  self->wait_ns = self->step_ns = self->n_synced = self->wake_ns = self->n_waited = 0;
  return r;
}

static PyObject* EnvLoop_f64_mode(EnvLoop* self, PyObject* Py_UNUSED(ignored)) {
  return PyBool_FromLong(self->f64_mode);
}

static PyMethodDef EnvLoop_methods[] = {
    {"step", (PyCFunction)EnvLoop_step, METH_VARARGS, "step(t, lazy): one time step of every env"},
    {"step_synced", (PyCFunction)EnvLoop_step_synced, METH_VARARGS,
     "step_synced(t, lazy, act_word, act_target, spin, obs_word, wake_at): wait, step, arrive"},
    {"timing", (PyCFunction)EnvLoop_timing, METH_NOARGS,
     "timing() -> (wait_ns, step_ns, calls) of step_synced since the last call (and reset)"},
    {"f64_mode", (PyCFunction)EnvLoop_f64_mode, METH_NOARGS, "True once the float64 sums are live"},
    {NULL, NULL, 0, NULL}};

static PyTypeObject EnvLoopType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "rlpyt_amd._envloop.EnvLoop",
    .tp_basicsize = sizeof(EnvLoop),
    .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_new = PyType_GenericNew,
    .tp_init = (initproc)EnvLoop_init,
    .tp_dealloc = (destructor)EnvLoop_dealloc,
    .tp_methods = EnvLoop_methods,
    .tp_doc = "Worker-side time step of a slice of environments (see csrc/envloop.c)",
};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_envloop",
                                    "C loop body of the sampler workers' env stepping", -1, NULL};

PyMODINIT_FUNC PyInit__envloop(void) {
  if (PyType_Ready(&EnvLoopType) < 0) return NULL;
  PyObject* m = PyModule_Create(&moddef);
  if (!m) return NULL;
  Py_INCREF(&EnvLoopType);
  if (PyModule_AddObject(m, "EnvLoop", (PyObject*)&EnvLoopType) < 0) {
    Py_DECREF(&EnvLoopType);
    Py_DECREF(m);
    return NULL;
  }
  return m;
}
