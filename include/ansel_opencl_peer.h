/* ansel_opencl_peer.h -- the names of src/common/opencl.h, bound to libansel_hip.
 *
 * The reference's host files on the export path -- src/develop/pixelpipe_gpu.c, src/develop/tiling.c,
 * src/caches/pixelpipe_cache.c, src/develop/blend.c -- and every module's process_cl() are written against
 * `common/opencl.h`.  A build that replaces src/common/opencl.c by libansel_hip.so installs THIS header as
 * common/opencl.h: each dt_opencl_*() the host files call is an inline forward to its dt_hip_*() peer with the same
 * arguments and the same return convention (TRUE/FALSE where the reference returns gboolean, CL_SUCCESS = 0 or a
 * negative code where it returns cl_int, NULL from the allocators).  tests/test_boundary_compile.py compiles the
 * reference's pixelpipe_gpu.c and tiling.c, unmodified, against this header and runs them on the device.
 *
 * What is NOT here, by design: the kernel table (dt_opencl_create_kernel / set_kernel_arg / enqueue_kernel_2d,
 * opencl.h:433-457).  Kernels are compiled into the library and reached through one entry point per module
 * (dt_hip_iop_<op>_process, ansel_hip.h section 2); a module's process_cl() becomes the four-line stub of
 * INTEGRATION.md section 2.  `cl_mem` is an opaque device handle here as it is there.
 */
#ifndef ANSEL_OPENCL_PEER_H
#define ANSEL_OPENCL_PEER_H

#include <stddef.h>
#include <stdint.h>

#include "ansel_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the handful of OpenCL names the host files use directly */
typedef void *cl_mem;
typedef int32_t cl_int;
typedef uint64_t cl_ulong;
#ifndef CL_SUCCESS
#define CL_SUCCESS 0
#define CL_TRUE 1
#define CL_FALSE 0
#define CL_MAP_READ (1 << 0)
#define CL_MAP_WRITE (1 << 1)
#define CL_MEM_READ_WRITE (1 << 0)
#endif
#define DT_OPENCL_DEFAULT_ERROR DT_HIP_DEFAULT_ERROR        /* opencl.h:61 */
#define DT_OPENCL_SYSMEM_ALLOCATION DT_HIP_SYSMEM_ALLOCATION /* opencl.h:62 */
#define ROUNDUPDWD(a, b) dt_opencl_dev_roundup_width(a, b)  /* opencl.h:85-86 */
#define ROUNDUPDHT(a, b) dt_opencl_dev_roundup_height(a, b)

/* opencl.h:168-175 */
typedef enum dt_opencl_fit_reason_t
{
  DT_OPENCL_FIT_OK = 0,
  DT_OPENCL_FIT_DIMENSION,
  DT_OPENCL_FIT_ALLOC_LIMIT,
  DT_OPENCL_FIT_AVAILABLE,
  DT_OPENCL_FIT_UNINITED
} dt_opencl_fit_reason_t;

/* ---- lifecycle, devices (opencl.h:333-336, 351-419, 460-470, 571-591) */
static inline int dt_opencl_is_inited(void) { return dt_hip_is_inited(); }
static inline int dt_opencl_is_enabled(void) { return dt_hip_is_enabled(); }
static inline int dt_opencl_update_settings(void) { return dt_hip_update_settings(); }
static inline int dt_opencl_get_num_devices(void) { return dt_hip_get_num_devices(); }
static inline const char *dt_opencl_get_device_name(const int devid) { return dt_hip_get_device_name(devid); }
static inline int dt_opencl_reserve_device_for_pipe(const int pipetype) { return dt_hip_reserve_device_for_pipe(pipetype); }
static inline void dt_opencl_reserve_device_by_id(const int devid) { dt_hip_reserve_device_by_id(devid); }
static inline int dt_opencl_try_reserve_device_by_id(const int devid) { return dt_hip_try_reserve_device_by_id(devid); }
static inline void dt_opencl_release_device(const int devid) { dt_hip_release_device(devid); }
static inline int dt_opencl_report_pipe_error(void) { return dt_hip_report_pipe_error(); }
static inline int dt_opencl_get_device_max_image_size(const int devid, int *width, int *height)
{
  return dt_hip_get_device_max_image_size(devid, width, height);
}
static inline size_t dt_opencl_get_device_max_global_mem(const int devid) { return dt_hip_get_device_max_global_mem(devid); }
static inline size_t dt_opencl_get_device_available(const int devid) { return dt_hip_get_device_available(devid); }
static inline size_t dt_opencl_get_device_memalloc(const int devid) { return dt_hip_get_device_memalloc(devid); }
static inline int dt_opencl_dev_roundup_width(int size, const int devid) { return dt_hip_dev_roundup_width(size, devid); }
static inline int dt_opencl_dev_roundup_height(int size, const int devid) { return dt_hip_dev_roundup_height(size, devid); }
static inline void dt_opencl_check_tuning(const int devid) { dt_hip_check_tuning(devid); }
static inline int dt_opencl_avoid_atomics(const int devid) { return dt_hip_avoid_atomics(devid); }
static inline int dt_opencl_micro_nap(const int devid) { return dt_hip_micro_nap(devid); }
static inline int dt_opencl_use_pinned_memory(const int devid) { return dt_hip_use_pinned_memory(devid); }
static inline int dt_opencl_image_fits_device(const int devid, const size_t width, const size_t height, const unsigned bpp,
                                              const float factor, const size_t overhead)
{
  return dt_hip_image_fits_device(devid, width, height, bpp, factor, overhead);
}
static inline dt_opencl_fit_reason_t dt_opencl_image_fits_device_reason(const int devid, const size_t width, const size_t height,
                                                                        const unsigned bpp, const float factor,
                                                                        const size_t overhead, size_t *needed, size_t *limit)
{
  if(!dt_hip_is_inited()) return DT_OPENCL_FIT_UNINITED;
  /* the peer has no per-dimension image limit: 0 fits, 1 = one buffer above the largest allocation, 2 = above free memory */
  const int r = dt_hip_image_fits_device_reason(devid, width, height, bpp, factor, overhead, needed, limit);
  return r == 0 ? DT_OPENCL_FIT_OK : (r == 1 ? DT_OPENCL_FIT_ALLOC_LIMIT : DT_OPENCL_FIT_AVAILABLE);
}

/* ---- memory (opencl.h:508-561) */
static inline void *dt_opencl_alloc_device(const int devid, const int width, const int height, const int bpp)
{
  return dt_hip_alloc_device(devid, width, height, bpp);
}
static inline void *dt_opencl_alloc_device_buffer(const int devid, const size_t size) { return dt_hip_alloc_device_buffer(devid, size); }
static inline void *dt_opencl_alloc_device_buffer_with_flags(const int devid, const size_t size, const int flags)
{
  (void)flags;
  return dt_hip_alloc_device_buffer(devid, size);
}
static inline void *dt_opencl_alloc_device_use_host_pointer(const int devid, const int width, const int height, const int bpp,
                                                            void *host, const int flags)
{
  return dt_hip_alloc_device_use_host_pointer(devid, width, height, bpp, host, flags);
}
static inline void dt_opencl_release_mem_object(cl_mem mem) { dt_hip_release_mem_object(mem); }
static inline size_t dt_opencl_get_mem_object_size(cl_mem mem) { return dt_hip_get_mem_object_size(mem); }
static inline int dt_opencl_get_mem_context_id(cl_mem mem) { return dt_hip_get_mem_context_id(mem); }
static inline int dt_opencl_get_image_width(cl_mem mem) { return dt_hip_get_image_width(mem); }
static inline int dt_opencl_get_image_height(cl_mem mem) { return dt_hip_get_image_height(mem); }
static inline int dt_opencl_get_image_element_size(cl_mem mem) { return dt_hip_get_image_element_size(mem); }
static inline void *dt_opencl_map_buffer(const int devid, cl_mem buffer, const int blocking, const int flags, size_t offset,
                                         size_t size)
{
  return dt_hip_map_buffer(devid, buffer, blocking, flags, offset, size);
}
static inline void *dt_opencl_map_image(const int devid, cl_mem img, const int blocking, const int flags, size_t width,
                                        size_t height, int bpp)
{
  return dt_hip_map_image(devid, img, blocking, flags, width, height, bpp);
}
static inline int dt_opencl_unmap_mem_object(const int devid, cl_mem mem, void *mapped_ptr)
{
  return dt_hip_unmap_mem_object(devid, mem, mapped_ptr);
}
static inline int dt_opencl_is_pinned_memory(cl_mem mem) { return dt_hip_is_pinned_memory(mem); }
static inline void dt_opencl_memory_statistics(int devid, cl_mem mem, int action)
{
  (void)devid;
  (void)mem;
  (void)action; /* the runtime keeps its own accounting: dt_hip_memory_statistics() */
}

/* ---- copies (opencl.h:473-537) */
static inline int dt_opencl_read_host_from_device(const int devid, void *host, void *device, const int width, const int height,
                                                  const int bpp)
{
  return dt_hip_read_host_from_device(devid, host, device, width, height, bpp);
}
static inline int dt_opencl_read_host_from_device_rowpitch(const int devid, void *host, void *device, const int width,
                                                           const int height, const int rowpitch)
{
  return dt_hip_read_host_from_device_rowpitch(devid, host, device, width, height, dt_hip_get_image_element_size(device),
                                               (size_t)rowpitch, 1);
}
static inline int dt_opencl_read_host_from_device_raw(const int devid, void *host, void *device, const size_t *origin,
                                                      const size_t *region, const int rowpitch, const int blocking)
{
  return dt_hip_read_host_from_device_raw(devid, host, device, origin, region, rowpitch, blocking);
}
static inline int dt_opencl_write_host_to_device(const int devid, void *host, void *device, const int width, const int height,
                                                 const int bpp)
{
  return dt_hip_write_host_to_device(devid, host, device, width, height, bpp);
}
static inline int dt_opencl_write_host_to_device_rowpitch(const int devid, void *host, void *device, const int width,
                                                          const int height, const int rowpitch)
{
  return dt_hip_write_host_to_device_rowpitch(devid, host, device, width, height, dt_hip_get_image_element_size(device),
                                              (size_t)rowpitch, 1);
}
static inline int dt_opencl_write_host_to_device_raw(const int devid, const void *host, void *device, const size_t *origin,
                                                     const size_t *region, const int rowpitch, const int blocking)
{
  return dt_hip_write_host_to_device_raw(devid, host, device, origin, region, rowpitch, blocking);
}
static inline void *dt_opencl_copy_host_to_device(const int devid, void *host, const int width, const int height, const int bpp)
{
  return dt_hip_copy_host_to_device(devid, host, width, height, bpp);
}
static inline void *dt_opencl_copy_host_to_device_rowpitch(const int devid, void *host, const int width, const int height,
                                                           const int bpp, const int rowpitch)
{
  return dt_hip_copy_host_to_device_rowpitch(devid, host, width, height, bpp, rowpitch);
}
static inline void *dt_opencl_copy_host_to_device_constant(const int devid, const size_t size, void *host)
{
  return dt_hip_copy_host_to_device_constant(devid, size, host);
}
static inline int dt_opencl_copy_device_to_host(const int devid, void *host, void *device, const int width, const int height,
                                                const int bpp)
{
  return dt_hip_copy_device_to_host(devid, host, device, width, height, bpp);
}
static inline int dt_opencl_enqueue_copy_image(const int devid, cl_mem src, cl_mem dst, size_t *orig_src, size_t *orig_dst,
                                               size_t *region)
{
  return dt_hip_enqueue_copy_image(devid, src, dst, orig_src, orig_dst, region);
}
static inline int dt_opencl_enqueue_copy_buffer_to_buffer(const int devid, cl_mem src, cl_mem dst, size_t srcoffset,
                                                          size_t dstoffset, size_t size)
{
  return dt_hip_enqueue_copy_buffer_to_buffer(devid, src, dst, srcoffset, dstoffset, size);
}
static inline int dt_opencl_read_buffer_from_device(const int devid, void *host, void *device, const size_t offset,
                                                    const size_t size, const int blocking)
{
  return dt_hip_read_buffer_from_device(devid, host, device, offset, size, blocking);
}
static inline int dt_opencl_write_buffer_to_device(const int devid, void *host, void *device, const size_t offset,
                                                   const size_t size, const int blocking)
{
  return dt_hip_write_buffer_to_device(devid, host, device, offset, size, blocking);
}

/* ---- sync, events (opencl.h:343-346, 594-608) */
static inline int dt_opencl_finish(const int devid) { return dt_hip_finish(devid); }
static inline int dt_opencl_enqueue_barrier(const int devid) { return dt_hip_enqueue_barrier(devid); }
static inline void dt_opencl_events_wait_for(const int devid) { dt_hip_events_wait_for(devid); }
static inline int dt_opencl_events_flush(const int devid, const int reset) { return dt_hip_events_flush(devid, reset); }
static inline void dt_opencl_events_reset(const int devid) { dt_hip_events_reset(devid); }

#ifdef __cplusplus
}
#endif
#endif /* ANSEL_OPENCL_PEER_H */
