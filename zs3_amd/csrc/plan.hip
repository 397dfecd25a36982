// zs3_plan_* / zs3_stream_wait (include/zs3hip.h, "recorded launch plans"): see plan.h for the design.
// Host code only; compiled by hipcc like the kernel sources so that the library stays one toolchain.
#include "plan.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "zs3hip.h"

namespace zs3 {

struct Op {
  int32_t fn;        // index into plan_fns, or FN_STREAM_WAIT
  uint32_t offset;   // of the argument block inside Plan::arena (8-byte aligned)
};
constexpr int32_t FN_STREAM_WAIT = -1;
struct WaitBlock {
  void* waiter;
  void* producer;
};

// HIP-event pairs around chosen ops of ONE replay (bench.py's roofline: the dominant kernel's launches timed inside replayed steps)
struct TimedSet {
  std::vector<int> ops;                 // ascending op indices
  std::vector<hipEvent_t> begin, end;   // one pair per op, recorded on the op's own stream
};

struct Plan {
  std::vector<Op> ops;
  std::vector<unsigned char> arena;
  std::mutex mu;          // forward launches come from the caller's thread, backward launches from autograd's device thread
  int failed_op = -1;
  int failed_rc = 0;
  std::vector<TimedSet*> timed;   // every set ever armed (read back by index)
  int armed = -1;                 // index of the set the NEXT replay records, -1 = none
  ~Plan() {
    for (TimedSet* t : timed) {
      for (hipEvent_t e : t->begin) hipEventDestroy(e);
      for (hipEvent_t e : t->end) hipEventDestroy(e);
      delete t;
    }
  }
};

static std::atomic<Plan*> g_recording{nullptr};

Plan* plan_recording() { return g_recording.load(std::memory_order_acquire); }

static void push_raw(Plan* plan, int32_t fn, const void* block, size_t bytes) {
  std::lock_guard<std::mutex> lock(plan->mu);
  const size_t off = (plan->arena.size() + 7) & ~size_t(7);
  plan->arena.resize(off + bytes);
  std::memcpy(plan->arena.data() + off, block, bytes);
  plan->ops.push_back(Op{fn, (uint32_t)off});
}

void plan_push(Plan* plan, int fn, const void* block) { push_raw(plan, fn, block, plan_fns[fn].block_bytes); }

// Events of the waits: a small ring per (waiting stream, producing stream) pair.  A wait refers to the record that precedes it, so
// an event can be recorded again for a later wait -- but not by leaning on that for back-to-back waits of ONE stream on several
// producers (the end-of-backward join: main waits for side 0, then side 1; lanes_join likewise): the first version kept one event
// per waiting stream and re-recorded it for the second producer microseconds after the first wait was enqueued, which is only
// correct if the runtime binds a wait to the event's state at enqueue time; a pair's ring of four makes no such assumption about
// any wait younger than three later waits of the same pair.
struct PairHash {
  size_t operator()(const std::pair<hipStream_t, hipStream_t>& k) const {
    return std::hash<void*>()((void*)k.first) * 1000003u ^ std::hash<void*>()((void*)k.second);
  }
};
struct EventRing {
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned next = 0;
};

static hipEvent_t wait_event_of(hipStream_t waiter, hipStream_t producer) {
  static std::mutex mu;
  static std::unordered_map<std::pair<hipStream_t, hipStream_t>, EventRing, PairHash> rings;
  std::lock_guard<std::mutex> lock(mu);
  EventRing& r = rings[std::make_pair(waiter, producer)];
  hipEvent_t& e = r.ev[r.next++ & 3];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}

static int stream_wait_impl(void* waiter, void* producer) {
  if (waiter == producer) return 0;
  hipEvent_t ev = wait_event_of((hipStream_t)waiter, (hipStream_t)producer);
  if (!ev) return (int)hipGetLastError();
  hipError_t rc = hipEventRecord(ev, (hipStream_t)producer);
  if (rc != hipSuccess) return (int)rc;
  return (int)hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
}

static Plan* as_plan(long handle) { return reinterpret_cast<Plan*>(handle); }

}  // namespace zs3

using namespace zs3;

extern "C" int zs3_stream_wait(void* waiter, void* producer) {
  if (Plan* pl = plan_recording()) {
    WaitBlock b{waiter, producer};
    push_raw(pl, FN_STREAM_WAIT, &b, sizeof b);
  }
  return stream_wait_impl(waiter, producer);
}

extern "C" long zs3_plan_create(void) { return reinterpret_cast<long>(new Plan()); }

extern "C" int zs3_plan_destroy(long plan) {
  Plan* pl = as_plan(plan);
  if (!pl) return -1;
  Plan* expected = pl;
  g_recording.compare_exchange_strong(expected, nullptr);
  delete pl;
  return 0;
}

extern "C" int zs3_plan_record_begin(long plan) {
  Plan* pl = as_plan(plan);
  if (!pl) return -1;
  Plan* expected = nullptr;
  if (!g_recording.compare_exchange_strong(expected, pl)) return expected == pl ? 0 : -2;   // another plan is recording
  return 0;
}

extern "C" int zs3_plan_record_end(long plan) {
  Plan* pl = as_plan(plan);
  if (!pl) return -1;
  Plan* expected = pl;
  if (!g_recording.compare_exchange_strong(expected, nullptr)) return -2;                    // this plan was not recording
  return (int)pl->ops.size();
}

extern "C" int zs3_plan_size(long plan) { return plan ? (int)as_plan(plan)->ops.size() : -1; }

extern "C" int zs3_plan_truncate(long plan, int nops) {
  Plan* pl = as_plan(plan);
  if (!pl || nops < 0 || nops > (int)pl->ops.size()) return -1;
  pl->ops.resize(nops);
  return 0;
}

extern "C" int zs3_plan_replay(long plan, int first, int count) {
  Plan* pl = as_plan(plan);
  if (!pl || first < 0) return -1;
  if (plan_recording() == pl) return -2;    // a plan cannot replay into itself
  const int n = (int)pl->ops.size();
  const int last = count < 0 ? n : (first + count < n ? first + count : n);
  const unsigned char* base = pl->arena.data();
  static const bool trace = getenv("ZS3_PLAN_TRACE") != nullptr;   // debugging: name every op on stderr and synchronise behind it
  TimedSet* ts = pl->armed >= 0 ? pl->timed[pl->armed] : nullptr;   // one-shot: this replay records the armed set's events
  pl->armed = -1;
  size_t tk = 0;
  for (int i = first; i < last; ++i) {
    const Op& op = pl->ops[i];
    hipStream_t tstream = nullptr;
    bool timed_op = false;
    if (ts) {
      while (tk < ts->ops.size() && ts->ops[tk] < i) ++tk;
      if (tk < ts->ops.size() && ts->ops[tk] == i && op.fn != FN_STREAM_WAIT) {
        const FnDesc& f = plan_fns[op.fn];
        std::memcpy(&tstream, base + op.offset + f.args[f.nargs - 1].offset, sizeof tstream);   // (the trailing `void* stream`)
        timed_op = true;
        hipEventRecord(ts->begin[tk], tstream);
      }
    }
    if (trace) {
      fprintf(stderr, "[zs3_plan] op %d %s\n", i, op.fn == FN_STREAM_WAIT ? "zs3_stream_wait" : plan_fns[op.fn].name);
      fflush(stderr);
    }
    int rc;
    if (op.fn == FN_STREAM_WAIT) {
      const WaitBlock* b = reinterpret_cast<const WaitBlock*>(base + op.offset);
      rc = stream_wait_impl(b->waiter, b->producer);
      if (Plan* rec = plan_recording()) push_raw(rec, FN_STREAM_WAIT, b, sizeof *b);   // (a replay inside another plan's recording)
    } else {
      rc = plan_fns[op.fn].call(base + op.offset);
      if (Plan* rec = plan_recording()) plan_push(rec, op.fn, base + op.offset);
    }
    if (timed_op) hipEventRecord(ts->end[tk], tstream);
    if (rc == 0 && trace) rc = (int)hipDeviceSynchronize();
    if (rc != 0) {
      pl->failed_op = i;
      pl->failed_rc = rc;
      return rc;
    }
  }
  return 0;
}

extern "C" int zs3_plan_time_ops(long plan, const int* ops, int n) {
  Plan* pl = as_plan(plan);
  if (!pl || n < 0 || (n > 0 && !ops)) return -1;
  if (n == 0) {
    pl->armed = -1;
    return 0;
  }
  TimedSet* t = new TimedSet();
  for (int k = 0; k < n; ++k) {
    if (ops[k] < 0 || ops[k] >= (int)pl->ops.size() || (k > 0 && ops[k] <= ops[k - 1]) || pl->ops[ops[k]].fn == FN_STREAM_WAIT) {
      delete t;
      return -3;
    }
    hipEvent_t b = nullptr, e = nullptr;
    if (hipEventCreate(&b) != hipSuccess || hipEventCreate(&e) != hipSuccess) {
      delete t;
      return (int)hipGetLastError();
    }
    t->ops.push_back(ops[k]);
    t->begin.push_back(b);
    t->end.push_back(e);
  }
  pl->timed.push_back(t);
  pl->armed = (int)pl->timed.size() - 1;
  return pl->armed;
}

extern "C" int zs3_plan_timed_ms(long plan, int set, float* out_ms, int cap) {
  Plan* pl = as_plan(plan);
  if (!pl || set < 0 || set >= (int)pl->timed.size() || !out_ms) return -1;
  TimedSet* t = pl->timed[set];
  const int n = (int)t->ops.size() < cap ? (int)t->ops.size() : cap;
  for (int k = 0; k < n; ++k) {
    const hipError_t rc = hipEventElapsedTime(&out_ms[k], t->begin[k], t->end[k]);
    if (rc != hipSuccess) return -(int)rc - 100;     // (not recorded / not finished: the caller synchronises first)
  }
  return n;
}

extern "C" int zs3_plan_failed_op(long plan) { return plan ? as_plan(plan)->failed_op : -1; }

extern "C" int zs3_plan_op_name(long plan, int op, char* buf, int cap) {
  Plan* pl = as_plan(plan);
  if (!pl || op < 0 || op >= (int)pl->ops.size() || !buf || cap <= 0) return -1;
  const int fn = pl->ops[op].fn;
  const char* name = fn == FN_STREAM_WAIT ? "zs3_stream_wait" : plan_fns[fn].name;
  std::strncpy(buf, name, (size_t)cap - 1);
  buf[cap - 1] = 0;
  return (int)std::strlen(name);
}

extern "C" int zs3_plan_find_op(long plan, const char* name, int nth) {
  Plan* pl = as_plan(plan);
  if (!pl || !name) return -1;
  for (int i = 0, seen = 0; i < (int)pl->ops.size(); ++i) {
    const int fn = pl->ops[i].fn;
    const char* nm = fn == FN_STREAM_WAIT ? "zs3_stream_wait" : plan_fns[fn].name;
    if (std::strcmp(nm, name) == 0 && seen++ == nth) return i;
  }
  return -1;
}

// every recorded argument of kind `kind` and width `bytes` that equals *old_value becomes *new_value; returns how many
static int replace_all(Plan* pl, const char* kinds, const void* old_value, const void* new_value, size_t bytes) {
  int hits = 0;
  unsigned char* base = pl->arena.data();
  for (const Op& op : pl->ops) {
    if (op.fn == FN_STREAM_WAIT) {
      if (std::strchr(kinds, 's') && bytes == sizeof(void*)) {
        WaitBlock* b = reinterpret_cast<WaitBlock*>(base + op.offset);
        for (void** s : {&b->waiter, &b->producer})
          if (std::memcmp(s, old_value, bytes) == 0) {
            std::memcpy(s, new_value, bytes);
            ++hits;
          }
      }
      continue;
    }
    const FnDesc& f = plan_fns[op.fn];
    for (int a = 0; a < f.nargs; ++a) {
      const ArgDesc& d = f.args[a];
      if (d.bytes != bytes || !std::strchr(kinds, d.kind)) continue;
      unsigned char* at = base + op.offset + d.offset;
      if (std::memcmp(at, old_value, bytes) == 0) {
        std::memcpy(at, new_value, bytes);
        ++hits;
      }
    }
  }
  return hits;
}

extern "C" int zs3_plan_replace_u64(long plan, unsigned long long old_value, unsigned long long new_value) {
  return plan ? replace_all(as_plan(plan), "u", &old_value, &new_value, sizeof old_value) : -1;
}

extern "C" int zs3_plan_replace_ptr(long plan, const void* old_ptr, const void* new_ptr) {
  return plan ? replace_all(as_plan(plan), "p", &old_ptr, &new_ptr, sizeof old_ptr) : -1;
}

extern "C" int zs3_plan_arg_kind(long plan, int op, int arg) {
  Plan* pl = as_plan(plan);
  if (!pl || op < 0 || op >= (int)pl->ops.size() || pl->ops[op].fn == FN_STREAM_WAIT) return -1;
  const FnDesc& f = plan_fns[pl->ops[op].fn];
  return arg < 0 || arg >= f.nargs ? 0 : (int)f.args[arg].kind;
}

extern "C" int zs3_plan_get_arg(long plan, int op, int arg, void* out, int cap) {
  Plan* pl = as_plan(plan);
  if (!pl || op < 0 || op >= (int)pl->ops.size() || pl->ops[op].fn == FN_STREAM_WAIT || !out) return -1;
  const FnDesc& f = plan_fns[pl->ops[op].fn];
  if (arg < 0 || arg >= f.nargs || cap < (int)f.args[arg].bytes) return -3;
  std::memcpy(out, pl->arena.data() + pl->ops[op].offset + f.args[arg].offset, f.args[arg].bytes);
  return (int)f.args[arg].bytes;
}

// where a device pointer occurs among the recorded arguments: up to cap (op, arg) pairs into where[2 * k], where[2 * k + 1]; -> total
extern "C" int zs3_plan_find_ptr(long plan, const void* ptr, int* where, int cap) {
  Plan* pl = as_plan(plan);
  if (!pl) return -1;
  int hits = 0;
  const unsigned char* base = pl->arena.data();
  for (int i = 0; i < (int)pl->ops.size(); ++i) {
    const Op& op = pl->ops[i];
    if (op.fn == FN_STREAM_WAIT) continue;
    const FnDesc& f = plan_fns[op.fn];
    for (int a = 0; a < f.nargs; ++a) {
      if (f.args[a].kind != 'p') continue;
      const void* v;
      std::memcpy(&v, base + op.offset + f.args[a].offset, sizeof v);
      if (v != ptr) continue;
      if (where && hits < cap) {
        where[2 * hits] = i;
        where[2 * hits + 1] = a;
      }
      ++hits;
    }
  }
  return hits;
}

extern "C" int zs3_plan_set_ptr(long plan, int op, int arg, const void* ptr) {
  Plan* pl = as_plan(plan);
  if (!pl || op < 0 || op >= (int)pl->ops.size() || pl->ops[op].fn == FN_STREAM_WAIT) return -1;
  const FnDesc& f = plan_fns[pl->ops[op].fn];
  if (arg < 0 || arg >= f.nargs) return -3;
  if (f.args[arg].kind != 'p') return -4;
  std::memcpy(pl->arena.data() + pl->ops[op].offset + f.args[arg].offset, &ptr, sizeof ptr);
  return 0;
}

extern "C" int zs3_plan_patch(long plan, int op, int arg, const void* data, int bytes) {
  Plan* pl = as_plan(plan);
  if (!pl || op < 0 || op >= (int)pl->ops.size() || pl->ops[op].fn == FN_STREAM_WAIT || !data) return -1;
  const FnDesc& f = plan_fns[pl->ops[op].fn];
  if (arg < 0 || arg >= f.nargs || bytes <= 0 || bytes > (int)f.args[arg].bytes) return -3;
  if (f.args[arg].kind == 'p' || f.args[arg].kind == 's') return -4;   // pointers and streams: zs3_plan_replace_ptr
  std::memcpy(pl->arena.data() + pl->ops[op].offset + f.args[arg].offset, data, (size_t)bytes);
  return 0;
}
