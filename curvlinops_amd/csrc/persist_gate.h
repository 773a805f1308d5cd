// Admission control for persistent grids (kernels whose workgroups wait for each other: csrc/mlp_mega.hip, the panel
// launches of csrc/sytrd.hip).  Such a grid only makes progress when ALL its workgroups are resident, and each of
// them holds a whole CU; two grids that are both PARTLY resident can therefore wait for each other forever.  The
// library never lets that happen: every persistent launch declares how many CUs it needs, and before it is queued its
// stream is made to wait (device side, hipStreamWaitEvent -- the host never blocks) for earlier persistent launches
// of OTHER streams until the CUs of everything still in flight plus its own fit the chip.  Launches of one stream are
// ordered by the stream itself.  State per device, guarded by a mutex; events are created once per stream and reused.
#pragma once
#include <algorithm>
#include <mutex>
#include <vector>

#include "clo_common.h"

namespace clo {

class PersistGate {
 public:
  // Call before queueing a persistent launch of `cus` workgroups (one per CU) on `st`; then launch; then done().
  int admit(hipStream_t st, int cus) {
    RelaxedCaptureScope relaxed;   // (event queries: see clo_common.h)
    mu_.lock();
    int total = cus;
    for (Slot &s : slots_) {
      if (!s.busy) continue;
      if (s.st == st) {   // own slot: forget the sizes of launches that have drained
        if (s.recorded && hipEventQuery(s.ev) == hipSuccess) s.busy = false;
        continue;
      }
      if (!s.recorded) {   // that stream was alone so far and skipped its records: an event NOW covers all it has queued
        if (hipEventRecord(s.ev, s.st) != hipSuccess) { mu_.unlock(); set_error("persistent launch: hipEventRecord failed"); return CLO_EHIP; }
        s.recorded = true;
      }
      if (hipEventQuery(s.ev) == hipSuccess) { s.busy = false; continue; }
      total += s.cus;
    }
    // make room: wait for the largest launches of other streams first
    while (total > cus_avail_) {
      Slot *big = nullptr;
      for (Slot &s : slots_)
        if (s.st != st && s.busy && !s.waited && (!big || s.cus > big->cus)) big = &s;
      if (!big) break;
      if (hipStreamWaitEvent(st, big->ev, 0) != hipSuccess) { mu_.unlock(); set_error("persistent launch: hipStreamWaitEvent failed"); return CLO_EHIP; }
      big->waited = true;
      total -= big->cus;
    }
    for (Slot &s : slots_) s.waited = false;
    pending_ = cus;
    return CLO_OK;   // (the mutex stays locked until done(): launch order == admission order)
  }
  int done(hipStream_t st) {
    RelaxedCaptureScope relaxed;
    Slot *mine = nullptr;
    for (Slot &s : slots_)
      if (s.st == st) mine = &s;
    int rc = CLO_OK;
    if (!mine) {
      Slot s;
      s.st = st;
      if (hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) { mu_.unlock(); set_error("persistent launch: hipEventCreate failed"); return CLO_EHIP; }
      slots_.push_back(s);
      mine = &slots_.back();
    }
    // a device with ONE launching stream needs no events (the stream orders its launches): the record is skipped
    // until a second stream shows up (admit() then records for it, which covers at least this launch)
    mine->recorded = false;
    if (slots_.size() >= 2) {
      if (hipEventRecord(mine->ev, st) != hipSuccess) { rc = CLO_EHIP; set_error("persistent launch: hipEventRecord failed"); }
      mine->recorded = true;
    }
    // the event covers EVERY launch queued on the stream so far: the slot carries the largest of those that may
    // still be in flight (busy is cleared only once an event of the stream has been seen complete)
    mine->cus = mine->busy ? std::max(pending_, mine->cus) : pending_;
    mine->busy = true;
    mu_.unlock();
    return rc;
  }
  void abort() { mu_.unlock(); }   // admit() succeeded but the launch did not happen
  static PersistGate &of(int dev) {
    static PersistGate gates[64];
    PersistGate &g = gates[dev & 63];
    if (g.cus_avail_ <= 0) g.cus_avail_ = device_cu_count(dev);   // (benign race: every writer stores the same value)
    return g;
  }
  int cus_available() const { return cus_avail_; }

 private:
  struct Slot { hipStream_t st = nullptr; hipEvent_t ev = nullptr; int cus = 0; bool busy = false, waited = false, recorded = false; };
  std::mutex mu_;
  std::vector<Slot> slots_;
  int pending_ = 0;
  int cus_avail_ = 0;   // compute units of the device (queried, not assumed)
};

}  // namespace clo
