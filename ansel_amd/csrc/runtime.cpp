// runtime.cpp -- device runtime of libansel_hip: the HIP peer of the reference's
// src/common/opencl.c (4.9 k lines of dlopen'd OpenCL: device discovery and priority
// opencl.c:1512-1640, per-device lock :1642-1755, image/buffer alloc :2593-2704, copies
// :2311-2455, event-list profiling :3048-3431, memory accounting :2783-2971).
//
// MI355X-first differences, on purpose:
//   * no program cache / kernel table / set_kernel_arg: kernels are compiled into this
//     library for gfx950 and launched directly by the module entry points;
//   * "images" are linear, tightly packed device buffers (the host cacheline layout), so a
//     module output can be handed to the next module, to a peer GPU or to the host without
//     a layout change;
//   * one in-order HIP stream per device replaces the OpenCL command queue; freed blocks go
//     to a size-keyed pool so a steady-state export performs no hipMalloc at all (288 GB of
//     HBM: we never need to give memory back between frames).
#include "hip_common.h"

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <map>
#include <vector>
#include <string>
#include <stdarg.h>

namespace
{

struct event_rec
{
  const char *tag;
  hipEvent_t start, stop;
};

struct device_t
{
  int hip_id = -1;
  std::string name;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::mutex lock; // device exclusivity: dt_opencl_reserve_device_for_pipe() / _release_device()
  bool events_enabled = false;
  std::mutex ev_lock; // events / event_pool: several host threads may launch on one device (dt_hip_pipe_process_bands)
  std::vector<event_rec> events;
  std::vector<hipEvent_t> event_pool;
  size_t cur_bytes = 0, peak_bytes = 0;
  std::multimap<size_t, void *> free_pool;
  // upload_small(): a ring of pinned staging buffers, each with the event behind its last copy
  struct staging_t
  {
    void *host = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool used = false;
  };
  std::mutex staging_lock;
  staging_t staging[8];
  int staging_next = 0;
};

struct alloc_t
{
  size_t size;
  int devid;
  int width, height, bpp; // dt_hip_alloc_device() geometry (0 for plain buffers)
  void *host;             // dt_hip_alloc_device_use_host_pointer(): the pinned host memory this aliases, not pooled
};

std::mutex g_mutex; // protects g_allocs, pools, counters
bool g_inited = false;
std::vector<device_t *> g_devs;
std::unordered_map<void *, alloc_t> g_allocs;
thread_local char t_error[512] = "";

} // namespace

namespace ansel
{

void set_last_error(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(t_error, sizeof(t_error), fmt, ap);
  va_end(ap);
}

bool valid_device(int devid) { return g_inited && devid >= 0 && devid < (int)g_devs.size(); }

// Streams, events and allocations belong to the device that is current when they are made, and HIP rejects an
// event recorded on a stream of another device.  The thread's current device is therefore made `devid`'s wherever
// work for `devid` starts: every module entry point fetches its stream through stream_of(), every allocation,
// event and stream creation goes through make_current().  A caller that drives several devices from one thread
// gets the right device per call; a thread per device (dt_hip_pipe_process_bands) pays one hipGetDevice() per call.
bool make_current(int devid)
{
  if(!valid_device(devid)) return false;
  const int want = g_devs[devid]->hip_id;
  int cur = -1;
  if(hipGetDevice(&cur) == hipSuccess && cur == want) return true;
  return hipSetDevice(want) == hipSuccess;
}

int hip_device_of(int devid) { return valid_device(devid) ? g_devs[devid]->hip_id : -1; }

// A module's small per-launch table (patch offsets, tile lists: what lived in a std::vector on the launcher's stack) -> device memory on
// the device's stream WITHOUT waiting for the stream: the bytes go through a pinned staging buffer of the device's ring, which is reused
// only behind the event recorded after its copy.  (The modules used to hipMemcpyAsync from the stack and then drain the stream -- a host
// meeting in the middle of a frame, which kept the host from enqueueing ahead and a batch's transfers from overlapping its kernels.)
int upload_small(int devid, void *dst_dev, const void *src_host, size_t bytes)
{
  if(!valid_device(devid) || !dst_dev || (!src_host && bytes)) return DT_HIP_INVALID_ARG;
  if(bytes == 0) return DT_HIP_SUCCESS;
  device_t *d = g_devs[devid];
  if(!make_current(devid)) return DT_HIP_DEFAULT_ERROR;
  std::lock_guard<std::mutex> g(d->staging_lock);
  device_t::staging_t &st = d->staging[d->staging_next];
  d->staging_next = (d->staging_next + 1) % 8;
  if(st.used && hipEventSynchronize(st.ev) != hipSuccess) return DT_HIP_DEFAULT_ERROR; // eight uploads ago: long done
  if(st.cap < bytes)
  {
    if(st.host) (void)hipHostFree(st.host);
    st.host = nullptr;
    st.cap = 0;
    size_t cap = 4096;
    while(cap < bytes) cap *= 2;
    if(hipHostMalloc(&st.host, cap, hipHostMallocDefault) != hipSuccess)
    {
      set_last_error("upload_small: no pinned staging buffer of %zu bytes", cap);
      st.host = nullptr;
      return DT_HIP_SYSMEM_ALLOCATION;
    }
    st.cap = cap;
  }
  if(!st.ev && hipEventCreateWithFlags(&st.ev, hipEventDisableTiming) != hipSuccess) return DT_HIP_DEFAULT_ERROR;
  memcpy(st.host, src_host, bytes);
  if(hipMemcpyAsync(dst_dev, st.host, bytes, hipMemcpyHostToDevice, d->stream) != hipSuccess || hipEventRecord(st.ev, d->stream) != hipSuccess)
  {
    set_last_error("upload_small: the copy of %zu bytes could not be enqueued", bytes);
    return DT_HIP_DEFAULT_ERROR;
  }
  st.used = true;
  return DT_HIP_SUCCESS;
}

hipStream_t stream_of(int devid)
{
  if(!make_current(devid)) return nullptr;
  return g_devs[devid]->stream;
}

launch_scope::launch_scope(int devid_, const char *tag_) : devid(devid_), tag(tag_), active(false)
{
  if(!valid_device(devid)) return;
  device_t *d = g_devs[devid];
  if(!d->events_enabled) return;
  make_current(devid); // an event belongs to the device that is current when it is created
  std::lock_guard<std::mutex> g(d->ev_lock);
  auto get = [&]() {
    hipEvent_t e;
    if(!d->event_pool.empty())
    {
      e = d->event_pool.back();
      d->event_pool.pop_back();
    }
    else if(hipEventCreate(&e) != hipSuccess)
      e = nullptr;
    return e;
  };
  start = get();
  stop = get();
  if(!start || !stop || hipEventRecord(start, d->stream) != hipSuccess)
  {
    // no timing for this launch; what was acquired goes back to the pool
    (void)hipGetLastError();
    if(start) d->event_pool.push_back(start);
    if(stop) d->event_pool.push_back(stop);
    return;
  }
  active = true;
}

launch_scope::~launch_scope()
{
  if(!active) return;
  device_t *d = g_devs[devid];
  std::lock_guard<std::mutex> g(d->ev_lock);
  if(hipEventRecord(stop, d->stream) != hipSuccess)
  {
    (void)hipGetLastError();
    d->event_pool.push_back(start);
    d->event_pool.push_back(stop);
    return;
  }
  d->events.push_back({ tag, start, stop });
}

} // namespace ansel

using namespace ansel;

extern "C" {

const char *dt_hip_last_error(void) { return t_error; }

int dt_hip_init(void)
{
  std::lock_guard<std::mutex> g(g_mutex);
  if(g_inited) return DT_HIP_SUCCESS;
  int n = 0;
  if(hipGetDeviceCount(&n) != hipSuccess || n <= 0)
  {
    set_last_error("dt_hip_init: no HIP device visible");
    return DT_HIP_DEFAULT_ERROR;
  }
  for(int i = 0; i < n; i++)
  {
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, i) != hipSuccess) continue;
    device_t *d = new device_t;
    d->hip_id = i;
    d->name = prop.name[0] ? prop.name : prop.gcnArchName;
    if(hipSetDevice(i) != hipSuccess || hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess)
    {
      delete d;
      continue;
    }
    d->own_stream = true;
    g_devs.push_back(d);
  }
  if(g_devs.empty())
  {
    set_last_error("dt_hip_init: no usable HIP device");
    return DT_HIP_DEFAULT_ERROR;
  }
  (void)hipSetDevice(g_devs[0]->hip_id);
  g_inited = true;
  return DT_HIP_SUCCESS;
}

void dt_hip_cleanup(void)
{
  std::lock_guard<std::mutex> g(g_mutex);
  if(!g_inited) return;
  for(device_t *d : g_devs)
  {
    (void)hipSetDevice(d->hip_id);
    (void)hipStreamSynchronize(d->stream);
    for(auto &kv : d->free_pool) (void)hipFree(kv.second);
    for(auto &e : d->events)
    {
      (void)hipEventDestroy(e.start);
      (void)hipEventDestroy(e.stop);
    }
    for(auto e : d->event_pool) (void)hipEventDestroy(e);
    for(auto &st : d->staging)
    {
      if(st.ev) (void)hipEventDestroy(st.ev);
      if(st.host) (void)hipHostFree(st.host);
    }
    if(d->own_stream) (void)hipStreamDestroy(d->stream);
    delete d;
  }
  // live allocations the caller never released; a device ALIAS of pinned host memory
  // (dt_hip_alloc_device_use_host_pointer) is not ours to free -- the host buffer stays with its owner
  for(auto &kv : g_allocs)
    if(!kv.second.host) (void)hipFree(kv.first);
  g_allocs.clear();
  g_devs.clear();
  g_inited = false;
}

int dt_hip_is_inited(void) { return g_inited ? 1 : 0; }

int dt_hip_get_num_devices(void) { return g_inited ? (int)g_devs.size() : 0; }

const char *dt_hip_get_device_name(int devid) { return valid_device(devid) ? g_devs[devid]->name.c_str() : ""; }

size_t dt_hip_get_device_available(int devid)
{
  if(!valid_device(devid)) return 0;
  size_t fr = 0, tot = 0;
  make_current(devid);
  if(hipMemGetInfo(&fr, &tot) != hipSuccess) return 0;
  // pooled blocks are ours to reuse
  std::lock_guard<std::mutex> g(g_mutex);
  for(auto &kv : g_devs[devid]->free_pool) fr += kv.first;
  return fr;
}

size_t dt_hip_get_device_memalloc(int devid)
{
  // HIP has no separate max-allocation limit below free memory on MI355X
  return dt_hip_get_device_available(devid);
}

// dt_opencl_reserve_device_for_pipe(), opencl.c:1642-1725: the first free device (one priority list: every device is
// an MI355X), -1 when all are busy.  Reserving a device and locking it are the same act.
int dt_hip_reserve_device_for_pipe(int pipetype)
{
  (void)pipetype;
  if(!g_inited) return -1;
  for(size_t i = 0; i < g_devs.size(); i++)
    if(g_devs[i]->lock.try_lock()) return (int)i;
  return -1;
}

// dt_opencl_reserve_device_by_id(), opencl.c:1737-1747: blocks until the device is free; out-of-range ids are ignored
void dt_hip_reserve_device_by_id(int devid)
{
  if(valid_device(devid)) g_devs[devid]->lock.lock();
}

// dt_opencl_try_reserve_device_by_id(), opencl.c:1749-1755: 0 when reserved (pthread convention), never waits
int dt_hip_try_reserve_device_by_id(int devid)
{
  if(!valid_device(devid)) return 1;
  return g_devs[devid]->lock.try_lock() ? 0 : 1;
}

// dt_opencl_release_device(), opencl.c:1727-1735
void dt_hip_release_device(int devid)
{
  if(valid_device(devid)) g_devs[devid]->lock.unlock();
}

// dt_opencl_get_device_max_image_size(), opencl.c:1772-1780.  Images are linear allocations here: what bounds a
// side is the 32-bit pixel coordinates of the kernels, not a texture unit
int dt_hip_get_device_max_image_size(int devid, int *width, int *height)
{
  if(!valid_device(devid)) return 0;
  if(width) *width = 1 << 20;
  if(height) *height = 1 << 20;
  return 1;
}

size_t dt_hip_get_device_max_global_mem(int devid)
{
  if(!valid_device(devid)) return 0;
  size_t free_b = 0, total_b = 0;
  make_current(devid);
  if(hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
  return total_b;
}

// dt_opencl_report_pipe_error(), opencl.c:1790-1802: 1 = this run failed, retry elsewhere; 2 = too many failures
// (DT_OPENCL_MAX_ERRORS 5, opencl.h:50), the device path is off for the session (dt_hip_is_enabled() -> 0)
static int g_error_count = 0;
static bool g_stopped = false;
int dt_hip_report_pipe_error(void)
{
  if(!g_inited) return 2;
  std::lock_guard<std::mutex> g(g_mutex);
  g_error_count++;
  if(g_error_count < 5) return 1;
  g_stopped = true;
  return 2;
}

// dt_opencl_is_enabled() / update_settings(), opencl.h:464-470
int dt_hip_is_enabled(void) { return (g_inited && !g_stopped) ? 1 : 0; }
int dt_hip_update_settings(void) { return dt_hip_is_enabled(); }

// per-device tuning knobs the host asks about (opencl.h:584-591, 648-651): nothing to tune on this part --
// no atomics workaround, no nap between launches, pinned transfers always
void dt_hip_check_tuning(int devid) { (void)devid; }
int dt_hip_avoid_atomics(int devid) { (void)devid; return 0; }
int dt_hip_micro_nap(int devid) { (void)devid; return 0; }
int dt_hip_use_pinned_memory(int devid) { return valid_device(devid) ? 1 : 0; }
// dt_opencl_dev_roundup_width / _height, opencl.c:2973-2982: images are linear allocations, nothing is rounded
int dt_hip_dev_roundup_width(int size, int devid) { (void)devid; return size; }
int dt_hip_dev_roundup_height(int size, int devid) { (void)devid; return size; }

int dt_hip_image_fits_device(int devid, size_t width, size_t height, unsigned bpp, float factor, size_t overhead)
{
  if(!valid_device(devid)) return 0;
  const double need = (double)width * (double)height * (double)bpp * (double)factor + (double)overhead;
  return need <= (double)dt_hip_get_device_available(devid) ? 1 : 0;
}

// dt_opencl_image_fits_device_reason(), opencl.h:577: 0 = fits, 1 = one buffer exceeds the largest allocation,
// 2 = the total exceeds the free memory; *needed / *limit report the pair that decided
int dt_hip_image_fits_device_reason(int devid, size_t width, size_t height, unsigned bpp, float factor, size_t overhead,
                                    size_t *needed, size_t *limit)
{
  if(!valid_device(devid)) return 2;
  const double single = (double)width * (double)height * (double)bpp;
  const double total = single * (double)factor + (double)overhead;
  const size_t avail = dt_hip_get_device_available(devid), memalloc = dt_hip_get_device_memalloc(devid);
  if(single > (double)memalloc)
  {
    if(needed) *needed = (size_t)single;
    if(limit) *limit = memalloc;
    return 1;
  }
  if(needed) *needed = (size_t)total;
  if(limit) *limit = avail;
  return total <= (double)avail ? 0 : 2;
}

void *dt_hip_get_stream(int devid) { return (void *)stream_of(devid); }

int dt_hip_set_stream(int devid, void *stream)
{
  if(!valid_device(devid)) return DT_HIP_INVALID_ARG;
  device_t *d = g_devs[devid];
  make_current(devid);
  if((hipStream_t)stream == d->stream) return DT_HIP_SUCCESS;
  // blocks released under the outgoing stream are handed out again under the new one: the pool's reuse is only
  // stream-ordered within ONE stream, so the outgoing stream is drained, owned or not
  if(d->stream || d->own_stream) (void)hipStreamSynchronize(d->stream);
  if(d->own_stream && d->stream) (void)hipStreamDestroy(d->stream);
  d->stream = (hipStream_t)stream;
  d->own_stream = false;
  return DT_HIP_SUCCESS;
}

dt_hip_mem_t dt_hip_alloc_device_buffer(int devid, size_t size)
{
  if(!valid_device(devid) || size == 0) return nullptr;
  device_t *d = g_devs[devid];
  const size_t rounded = (size + 255) & ~(size_t)255;
  std::lock_guard<std::mutex> g(g_mutex);
  void *p = nullptr;
  auto it = d->free_pool.find(rounded);
  if(it != d->free_pool.end())
  {
    p = it->second;
    d->free_pool.erase(it);
  }
  else
  {
    make_current(devid);
    hipError_t e = hipMalloc(&p, rounded);
    if(e != hipSuccess)
    {
      // flush the pool and retry once: dt_opencl_alloc_device's "flush cached cl_mem and
      // retry" (src/common/opencl.c:2604-2616)
      for(auto &kv : d->free_pool) (void)hipFree(kv.second);
      d->free_pool.clear();
      (void)hipGetLastError();
      e = hipMalloc(&p, rounded);
    }
    if(e != hipSuccess)
    {
      set_last_error("dt_hip_alloc_device_buffer: %zu bytes: %s", rounded, hipGetErrorString(e));
      (void)hipGetLastError();
      return nullptr;
    }
  }
  g_allocs[p] = { rounded, devid, 0, 0, 0, nullptr };
  d->cur_bytes += rounded;
  if(d->cur_bytes > d->peak_bytes) d->peak_bytes = d->cur_bytes;
  return p;
}

dt_hip_mem_t dt_hip_alloc_device(int devid, int width, int height, int bpp)
{
  if(width <= 0 || height <= 0 || bpp <= 0) return nullptr;
  dt_hip_mem_t p = dt_hip_alloc_device_buffer(devid, (size_t)width * (size_t)height * (size_t)bpp);
  if(p)
  {
    std::lock_guard<std::mutex> g(g_mutex);
    alloc_t &a = g_allocs[p];
    a.width = width;
    a.height = height;
    a.bpp = bpp;
  }
  return p;
}

void dt_hip_release_mem_object(dt_hip_mem_t mem)
{
  if(!mem) return;
  std::lock_guard<std::mutex> g(g_mutex);
  auto it = g_allocs.find(mem);
  if(it == g_allocs.end()) return; // not ours (e.g. a torch tensor): nothing to do
  const alloc_t a = it->second;
  g_allocs.erase(it);
  if(a.host) return; // an alias of pinned host memory: the host buffer stays with its owner
  if(a.devid >= 0 && a.devid < (int)g_devs.size())
  {
    device_t *d = g_devs[a.devid];
    d->cur_bytes -= a.size;
    // stream-ordered reuse is safe: every consumer of this runtime enqueues on d->stream
    d->free_pool.insert({ a.size, mem });
  }
  else
    (void)hipFree(mem);
}

size_t dt_hip_get_mem_object_size(dt_hip_mem_t mem)
{
  std::lock_guard<std::mutex> g(g_mutex);
  auto it = g_allocs.find(mem);
  return it == g_allocs.end() ? 0 : it->second.size;
}

// dt_opencl_get_image_width / _height / _element_size, opencl.h:554-558
static alloc_t alloc_of(dt_hip_mem_t mem)
{
  std::lock_guard<std::mutex> g(g_mutex);
  auto it = g_allocs.find(mem);
  return it == g_allocs.end() ? alloc_t{ 0, -1, 0, 0, 0, nullptr } : it->second;
}
int dt_hip_get_image_width(dt_hip_mem_t mem) { return alloc_of(mem).width; }
int dt_hip_get_image_height(dt_hip_mem_t mem) { return alloc_of(mem).height; }
int dt_hip_get_image_element_size(dt_hip_mem_t mem) { return alloc_of(mem).bpp; }
int dt_hip_get_mem_context_id(dt_hip_mem_t mem) { return alloc_of(mem).devid; }

// dt_opencl_alloc_device_use_host_pointer(), opencl.h:521: a device view of page-locked host memory (zero copy);
// NULL unless `host` is pinned (dt_hip_alloc_host_pinned), in which case the caller takes the copying path
dt_hip_mem_t dt_hip_alloc_device_use_host_pointer(int devid, int width, int height, int bpp, void *host, int flags)
{
  (void)flags;
  if(!valid_device(devid) || !host || width <= 0 || height <= 0 || bpp <= 0 || !dt_hip_is_pinned_memory(host)) return nullptr;
  void *dev = nullptr;
  make_current(devid);
  if(hipHostGetDevicePointer(&dev, host, 0) != hipSuccess || !dev)
  {
    (void)hipGetLastError();
    return nullptr;
  }
  std::lock_guard<std::mutex> g(g_mutex);
  g_allocs[dev] = { (size_t)width * height * bpp, devid, width, height, bpp, host };
  return dev;
}

// dt_opencl_map_image / map_buffer / unmap_mem_object, opencl.h:545-550 (pixelpipe_cache.c maps pinned cache lines):
// an object made by dt_hip_alloc_device_use_host_pointer() maps to its host memory once the stream has drained;
// device-only memory has no host mapping (NULL: the caller copies)
void *dt_hip_map_buffer(int devid, dt_hip_mem_t buffer, int blocking, int flags, size_t offset, size_t size)
{
  (void)flags;
  (void)size;
  const alloc_t a = alloc_of(buffer);
  if(!a.host || !valid_device(devid)) return nullptr;
  if(blocking && hipStreamSynchronize(stream_of(devid)) != hipSuccess) return nullptr;
  return (char *)a.host + offset;
}
void *dt_hip_map_image(int devid, dt_hip_mem_t buffer, int blocking, int flags, size_t width, size_t height, int bpp)
{
  (void)width;
  (void)height;
  (void)bpp;
  return dt_hip_map_buffer(devid, buffer, blocking, flags, 0, 0);
}
int dt_hip_unmap_mem_object(int devid, dt_hip_mem_t mem, void *mapped_ptr)
{
  (void)mapped_ptr;
  return (valid_device(devid) && alloc_of(mem).host) ? DT_HIP_SUCCESS : DT_HIP_INVALID_ARG;
}

// dt_opencl_copy_host_to_device[_rowpitch|_constant], opencl.h:508-514: allocate + upload, NULL on failure
dt_hip_mem_t dt_hip_copy_host_to_device_rowpitch(int devid, void *host, int width, int height, int bpp, int rowpitch)
{
  dt_hip_mem_t d = dt_hip_alloc_device(devid, width, height, bpp);
  if(!d) return nullptr;
  if(dt_hip_write_host_to_device_rowpitch(devid, host, d, width, height, bpp, (size_t)rowpitch, 1) != DT_HIP_SUCCESS)
  {
    dt_hip_release_mem_object(d);
    return nullptr;
  }
  return d;
}
dt_hip_mem_t dt_hip_copy_host_to_device(int devid, void *host, int width, int height, int bpp)
{
  return dt_hip_copy_host_to_device_rowpitch(devid, host, width, height, bpp, 0);
}
dt_hip_mem_t dt_hip_copy_host_to_device_constant(int devid, size_t size, void *host)
{
  dt_hip_mem_t d = dt_hip_alloc_device_buffer(devid, size);
  if(!d) return nullptr;
  if(dt_hip_write_buffer_to_device(devid, host, d, 0, size, 1) != DT_HIP_SUCCESS)
  {
    dt_hip_release_mem_object(d);
    return nullptr;
  }
  return d;
}
int dt_hip_copy_device_to_host(int devid, void *host, dt_hip_mem_t device, int width, int height, int bpp)
{
  return dt_hip_read_host_from_device(devid, host, device, width, height, bpp);
}

// dt_opencl_read_buffer_from_device / write_buffer_to_device, opencl.h:533-537
int dt_hip_read_buffer_from_device(int devid, void *host, dt_hip_mem_t device, size_t offset, size_t size, int blocking)
{
  if(!valid_device(devid) || !host || !device) return DT_HIP_INVALID_ARG;
  if(!size) return DT_HIP_SUCCESS;
  hipStream_t s = stream_of(devid);
  ANSEL_HIP_CHECK(hipMemcpyAsync(host, (const char *)device + offset, size, hipMemcpyDeviceToHost, s));
  if(blocking) ANSEL_HIP_CHECK(hipStreamSynchronize(s));
  return DT_HIP_SUCCESS;
}
int dt_hip_write_buffer_to_device(int devid, const void *host, dt_hip_mem_t device, size_t offset, size_t size, int blocking)
{
  if(!valid_device(devid) || !host || !device) return DT_HIP_INVALID_ARG;
  if(!size) return DT_HIP_SUCCESS;
  hipStream_t s = stream_of(devid);
  ANSEL_HIP_CHECK(hipMemcpyAsync((char *)device + offset, host, size, hipMemcpyHostToDevice, s));
  // a pageable source is staged before hipMemcpyAsync returns, a pinned one is read when the copy runs
  if(blocking || dt_hip_is_pinned_memory(host)) ANSEL_HIP_CHECK(hipStreamSynchronize(s));
  return DT_HIP_SUCCESS;
}

// dt_opencl_enqueue_barrier(), opencl.h:346: the device's stream is in order, every launch is already a barrier
int dt_hip_enqueue_barrier(int devid) { return valid_device(devid) ? DT_HIP_SUCCESS : DT_HIP_INVALID_ARG; }

void dt_hip_memory_statistics(int devid, size_t *current, size_t *peak)
{
  if(current) *current = valid_device(devid) ? g_devs[devid]->cur_bytes : 0;
  if(peak) *peak = valid_device(devid) ? g_devs[devid]->peak_bytes : 0;
}

int dt_hip_write_host_to_device_rowpitch(int devid, const void *host, dt_hip_mem_t device, int width, int height,
                                         int bpp, size_t rowpitch, int blocking)
{
  if(!valid_device(devid) || !host || !device) return DT_HIP_INVALID_ARG;
  hipStream_t s = stream_of(devid);
  const size_t wbytes = (size_t)width * bpp;
  launch_scope ls(devid, "[Write Image (from host to device)]");
  if(rowpitch == wbytes || rowpitch == 0) // 0 = tightly packed, the OpenCL convention
    ANSEL_HIP_CHECK(hipMemcpyAsync(device, host, wbytes * height, hipMemcpyHostToDevice, s));
  else
    ANSEL_HIP_CHECK(hipMemcpy2DAsync(device, wbytes, host, rowpitch, wbytes, height, hipMemcpyHostToDevice, s));
  if(blocking) ANSEL_HIP_CHECK(hipStreamSynchronize(s));
  return DT_HIP_SUCCESS;
}

int dt_hip_write_host_to_device(int devid, const void *host, dt_hip_mem_t device, int width, int height, int bpp)
{
  return dt_hip_write_host_to_device_rowpitch(devid, host, device, width, height, bpp, (size_t)width * bpp, 1);
}

int dt_hip_read_host_from_device_rowpitch(int devid, void *host, dt_hip_mem_t device, int width, int height,
                                          int bpp, size_t rowpitch, int blocking)
{
  if(!valid_device(devid) || !host || !device) return DT_HIP_INVALID_ARG;
  hipStream_t s = stream_of(devid);
  const size_t wbytes = (size_t)width * bpp;
  launch_scope ls(devid, "[Read Image (from device to host)]");
  if(rowpitch == wbytes || rowpitch == 0) // 0 = tightly packed, the OpenCL convention
    ANSEL_HIP_CHECK(hipMemcpyAsync(host, device, wbytes * height, hipMemcpyDeviceToHost, s));
  else
    ANSEL_HIP_CHECK(hipMemcpy2DAsync(host, rowpitch, device, wbytes, wbytes, height, hipMemcpyDeviceToHost, s));
  if(blocking) ANSEL_HIP_CHECK(hipStreamSynchronize(s));
  return DT_HIP_SUCCESS;
}

int dt_hip_read_host_from_device(int devid, void *host, dt_hip_mem_t device, int width, int height, int bpp)
{
  return dt_hip_read_host_from_device_rowpitch(devid, host, device, width, height, bpp, (size_t)width * bpp, 1);
}

// dt_opencl_read_host_from_device_raw / write_host_to_device_raw (opencl.h:489-506): a window {origin, region} of a
// 2-D device image against host memory with its own row pitch -- what the host tilers move per tile (tiling.c:1018,
// 1345).  The image's width and pixel size are the ones dt_hip_alloc_device() recorded.
static int raw_window(int devid, dt_hip_mem_t device, const size_t *origin, const size_t *region, size_t &wbytes, size_t &dpitch,
                      size_t &doffs)
{
  if(!valid_device(devid) || !device || !origin || !region) return DT_HIP_INVALID_ARG;
  const alloc_t a = alloc_of(device);
  if(a.width <= 0 || a.bpp <= 0)
  {
    set_last_error("window copy: the device object is not a 2-D image of this runtime (dt_hip_alloc_device)");
    return DT_HIP_INVALID_ARG;
  }
  if(origin[0] + region[0] > (size_t)a.width || origin[1] + region[1] > (size_t)a.height) return DT_HIP_INVALID_ARG;
  wbytes = region[0] * a.bpp;
  dpitch = (size_t)a.width * a.bpp;
  doffs = origin[1] * dpitch + origin[0] * a.bpp;
  return DT_HIP_SUCCESS;
}

int dt_hip_read_host_from_device_raw(int devid, void *host, dt_hip_mem_t device, const size_t *origin, const size_t *region,
                                     int rowpitch, int blocking)
{
  size_t wbytes, dpitch, doffs;
  const int rc = raw_window(devid, device, origin, region, wbytes, dpitch, doffs);
  if(rc != DT_HIP_SUCCESS || !host) return rc != DT_HIP_SUCCESS ? rc : DT_HIP_INVALID_ARG;
  if(!wbytes || !region[1]) return DT_HIP_SUCCESS;
  hipStream_t s = stream_of(devid);
  launch_scope ls(devid, "[Read Image (from device to host)]");
  ANSEL_HIP_CHECK(hipMemcpy2DAsync(host, rowpitch ? (size_t)rowpitch : wbytes, (const char *)device + doffs, dpitch, wbytes,
                                   region[1], hipMemcpyDeviceToHost, s));
  if(blocking) ANSEL_HIP_CHECK(hipStreamSynchronize(s));
  return DT_HIP_SUCCESS;
}

int dt_hip_write_host_to_device_raw(int devid, const void *host, dt_hip_mem_t device, const size_t *origin, const size_t *region,
                                    int rowpitch, int blocking)
{
  size_t wbytes, dpitch, doffs;
  const int rc = raw_window(devid, device, origin, region, wbytes, dpitch, doffs);
  if(rc != DT_HIP_SUCCESS || !host) return rc != DT_HIP_SUCCESS ? rc : DT_HIP_INVALID_ARG;
  if(!wbytes || !region[1]) return DT_HIP_SUCCESS;
  hipStream_t s = stream_of(devid);
  launch_scope ls(devid, "[Write Image (from host to device)]");
  ANSEL_HIP_CHECK(hipMemcpy2DAsync((char *)device + doffs, dpitch, host, rowpitch ? (size_t)rowpitch : wbytes, wbytes, region[1],
                                   hipMemcpyHostToDevice, s));
  // a pageable source is staged before the call returns, a pinned one is read when the copy runs
  if(blocking || dt_hip_is_pinned_memory(host)) ANSEL_HIP_CHECK(hipStreamSynchronize(s));
  return DT_HIP_SUCCESS;
}

// dt_opencl_enqueue_copy_image (opencl.h:516): a window between two 2-D images of equal pixel size
int dt_hip_enqueue_copy_image(int devid, dt_hip_mem_t src, dt_hip_mem_t dst, const size_t *orig_src, const size_t *orig_dst,
                              const size_t *region)
{
  if(!valid_device(devid) || !src || !dst || !orig_src || !orig_dst || !region) return DT_HIP_INVALID_ARG;
  const alloc_t a = alloc_of(src), b = alloc_of(dst);
  if(a.width <= 0 || b.width <= 0 || a.bpp != b.bpp) return DT_HIP_INVALID_ARG;
  // clEnqueueCopyImage returns CL_INVALID_VALUE for a window that leaves either image
  const size_t lim = 0x7fffffff;
  if(region[0] > lim || region[1] > lim || orig_src[0] > lim || orig_src[1] > lim || orig_dst[0] > lim || orig_dst[1] > lim
     || orig_src[0] + region[0] > (size_t)a.width || orig_dst[0] + region[0] > (size_t)b.width
     || (a.height > 0 && orig_src[1] + region[1] > (size_t)a.height) || (b.height > 0 && orig_dst[1] + region[1] > (size_t)b.height))
  {
    set_last_error("enqueue_copy_image: the window leaves the image");
    return DT_HIP_INVALID_ARG;
  }
  return dt_hip_enqueue_copy_region(devid, src, a.width, (int)orig_src[0], (int)orig_src[1], dst, b.width, (int)orig_dst[0],
                                    (int)orig_dst[1], (int)region[0], (int)region[1], a.bpp);
}

void *dt_hip_alloc_host_pinned(size_t size)
{
  void *p = NULL;
  if(size == 0 || hipHostMalloc(&p, size, hipHostMallocDefault) != hipSuccess)
  {
    (void)hipGetLastError();
    return NULL;
  }
  return p;
}

void dt_hip_free_host_pinned(void *host)
{
  if(host) (void)hipHostFree(host);
}

int dt_hip_is_pinned_memory(const void *host)
{
  hipPointerAttribute_t attr;
  if(!host || hipPointerGetAttributes(&attr, host) != hipSuccess)
  {
    (void)hipGetLastError();
    return 0;
  }
  return attr.type == hipMemoryTypeHost;
}

// process() of iop/basebuffer.c:118-160: crop-copy of the sensor buffer into the first cacheline
int dt_hip_iop_basebuffer_process(int devid, const dt_hip_piece_t *piece, int iwidth, int iheight, int bpp,
                                  const void *host_full, dt_hip_mem_t dev_out)
{
  if(!valid_device(devid) || !piece || !host_full || !dev_out || iwidth <= 0 || iheight <= 0 || bpp <= 0)
    return DT_HIP_INVALID_ARG;
  const size_t x = piece->roi_out.x > 0 ? (size_t)piece->roi_out.x : 0, y = piece->roi_out.y > 0 ? (size_t)piece->roi_out.y : 0;
  if(piece->roi_out.width <= 0 || piece->roi_out.height <= 0 || x >= (size_t)iwidth || y >= (size_t)iheight)
    return DT_HIP_SUCCESS;
  const size_t in_width = std::min((size_t)piece->roi_out.width, (size_t)iwidth - x);
  const size_t in_height = std::min((size_t)piece->roi_out.height, (size_t)iheight - y);
  const size_t in_stride = (size_t)iwidth * bpp, out_stride = (size_t)piece->roi_out.width * bpp;
  const size_t row_bytes = std::min(in_width * bpp, out_stride);
  const char *src = (const char *)host_full + y * in_stride + x * bpp;
  hipStream_t s = stream_of(devid);
  launch_scope ls(devid, "[Write Image (from host to device)]");
  ANSEL_HIP_CHECK(hipMemcpy2DAsync(dev_out, out_stride, src, in_stride, row_bytes, in_height, hipMemcpyHostToDevice, s));
  ANSEL_HIP_CHECK(hipStreamSynchronize(s)); // the source is the caller's memory
  return DT_HIP_SUCCESS;
}

int dt_hip_enqueue_copy_buffer_to_buffer(int devid, dt_hip_mem_t src, dt_hip_mem_t dst, size_t srcoffset,
                                         size_t dstoffset, size_t size)
{
  if(!valid_device(devid) || !src || !dst) return DT_HIP_INVALID_ARG;
  launch_scope ls(devid, "[Copy Buffer to Buffer (on device)]");
  ANSEL_HIP_CHECK(hipMemcpyAsync((char *)dst + dstoffset, (const char *)src + srcoffset, size,
                                 hipMemcpyDeviceToDevice, stream_of(devid)));
  return DT_HIP_SUCCESS;
}

int dt_hip_enqueue_copy_region(int devid, dt_hip_mem_t src, int src_width, int src_x, int src_y, dt_hip_mem_t dst,
                               int dst_width, int dst_x, int dst_y, int width, int height, int bpp)
{
  if(!valid_device(devid) || !src || !dst) return DT_HIP_INVALID_ARG;
  if(width <= 0 || height <= 0) return DT_HIP_SUCCESS;
  launch_scope ls(devid, "[Copy Image (on device)]");
  const char *s = (const char *)src + ((size_t)src_y * src_width + src_x) * bpp;
  char *d = (char *)dst + ((size_t)dst_y * dst_width + dst_x) * bpp;
  ANSEL_HIP_CHECK(hipMemcpy2DAsync(d, (size_t)dst_width * bpp, s, (size_t)src_width * bpp, (size_t)width * bpp,
                                   height, hipMemcpyDeviceToDevice, stream_of(devid)));
  return DT_HIP_SUCCESS;
}

int dt_hip_finish(int devid)
{
  if(!valid_device(devid)) return 0;
  const hipError_t e = hipStreamSynchronize(stream_of(devid));
  if(e != hipSuccess)
  {
    set_last_error("dt_hip_finish: %s", hipGetErrorString(e));
    return 0;
  }
  return 1; // TRUE on success, like dt_opencl_finish
}

void dt_hip_events_enable(int devid, int enable)
{
  if(valid_device(devid)) g_devs[devid]->events_enabled = enable != 0;
}

void dt_hip_events_reset(int devid)
{
  if(!valid_device(devid)) return;
  device_t *d = g_devs[devid];
  (void)hipStreamSynchronize(d->stream);
  for(auto &e : d->events)
  {
    d->event_pool.push_back(e.start);
    d->event_pool.push_back(e.stop);
  }
  d->events.clear();
}

// dt_opencl_events_wait_for() / events_flush(), opencl.h:601-605: the tagged launches live on one in-order stream,
// so waiting for them is draining it; flush reports the stream's status and optionally drops the records
void dt_hip_events_wait_for(int devid)
{
  if(valid_device(devid)) (void)hipStreamSynchronize(g_devs[devid]->stream);
}

int dt_hip_events_flush(int devid, int reset)
{
  if(!valid_device(devid)) return DT_HIP_INVALID_ARG;
  const hipError_t e = hipStreamSynchronize(g_devs[devid]->stream);
  if(reset) dt_hip_events_reset(devid);
  if(e != hipSuccess)
  {
    set_last_error("dt_hip_events_flush: %s", hipGetErrorString(e));
    return DT_HIP_DEFAULT_ERROR;
  }
  return DT_HIP_SUCCESS;
}

int dt_hip_events_profiling(int devid, const char **tags, float *ms, int *counts, int max)
{
  if(!valid_device(devid)) return 0;
  device_t *d = g_devs[devid];
  (void)hipStreamSynchronize(d->stream);
  std::vector<const char *> order;
  std::unordered_map<std::string, int> idx;
  std::vector<float> tsum;
  std::vector<int> cnt;
  for(auto &e : d->events)
  {
    float t = 0.f;
    if(hipEventElapsedTime(&t, e.start, e.stop) != hipSuccess) continue;
    auto it = idx.find(e.tag);
    int k;
    if(it == idx.end())
    {
      k = (int)order.size();
      idx[e.tag] = k;
      order.push_back(e.tag);
      tsum.push_back(0.f);
      cnt.push_back(0);
    }
    else
      k = it->second;
    tsum[k] += t;
    cnt[k] += 1;
  }
  const int n = (int)order.size();
  for(int i = 0; i < n && i < max; i++)
  {
    if(tags) tags[i] = order[i];
    if(ms) ms[i] = tsum[i];
    if(counts) counts[i] = cnt[i];
  }
  return n;
}

// sizes of the C-ABI structs as this library was compiled, for the ctypes mirror's self-check
size_t dt_hip_abi_sizeof(const char *name)
{
#define S(n, t) \
  if(!strcmp(name, n)) return sizeof(t)
  S("roi", dt_hip_roi_t);
  S("piece", dt_hip_piece_t);
  S("tiling", dt_hip_tiling_t);
  S("rawprepare", dt_hip_rawprepare_data_t);
  S("temperature", dt_hip_temperature_data_t);
  S("highlights", dt_hip_highlights_data_t);
  S("demosaic", dt_hip_demosaic_data_t);
  S("exposure", dt_hip_exposure_data_t);
  S("conversion", dt_hip_conversion_t);
  S("channelmixerrgb", dt_hip_channelmixerrgb_data_t);
  S("filmic_spline", dt_hip_filmic_spline_t);
  S("filmicrgb", dt_hip_filmicrgb_data_t);
  S("diffuse", dt_hip_diffuse_data_t);
  S("denoiseprofile", dt_hip_denoiseprofile_data_t);
  S("nlmeans", dt_hip_nlmeans_data_t);
  S("lab", dt_hip_lab_data_t);
  S("bilat", dt_hip_bilat_data_t);
  S("finalscale", dt_hip_finalscale_data_t);
  S("blend", dt_hip_blend_data_t);
  S("detailmask", dt_hip_detailmask_data_t);
  S("export_rows", dt_hip_export_rows_t);
  S("tile_plan", dt_hip_tile_plan_t);
  S("band", dt_hip_band_t);
  S("band_state", dt_hip_band_state_t);
  S("band_stats", dt_hip_band_stats_t);
#undef S
  return 0;
}

} // extern "C"
