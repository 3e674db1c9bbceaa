/* oracle/ref_wrap/ref_runtime.c -- TEST INFRASTRUCTURE ONLY.
 * Allocation/threads runtime the reference pixel code links against
 * (stand-ins for src/caches/pixelpipe_cache.c arena and src/system/mem_alloc.c). */
#include "ref_host.h"

int dt_get_num_openmp_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void *dt_alloc_align(size_t size)
{
  void *p = NULL;
  if(posix_memalign(&p, 64, size ? size : 64)) return NULL;
  /* Scratch starts zeroed: the reference reads a few scratch words it never wrote (see
   * oracle/src/demosaic_rcd.c), and a recycled heap block would make those reads depend on
   * what this process computed before. */
  memset(p, 0, size ? size : 64);
  return p;
}

void *dt_pixelpipe_cache_alloc_align_cache_impl(size_t size, int id, const char *name)
{
  (void)id; (void)name;
  return dt_alloc_align(size);
}

void dt_pixelpipe_cache_free_align_cache(void **mem, const char *message)
{
  (void)message;
  if(mem && *mem) { free(*mem); *mem = NULL; }
}

/* glib's allocator pair, so that the few g_malloc/g_free sites of the lifted code
 * (src/math/gaussian_elimination.h) link without libglib */
void *g_malloc(gsize n) { return n ? malloc(n) : NULL; }
void *g_malloc0(gsize n) { return n ? calloc(1, n) : NULL; }
void g_free(void *p) { free(p); }

void ref_set_num_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

int ref_get_num_threads(void) { return dt_get_num_openmp_threads(); }

/* The reference leaves FTZ/DAZ set in every OpenMP worker after RCD ran
 * (src/iop/demosaic/rcd.c:300 never restores MXCSR).  To keep each wrapped call's
 * arithmetic independent of call history we force the default IEEE mode on every
 * worker at the entry of each wrapper; rcd_demosaic() then sets its own fast mode. */
void ref_reset_fp_mode(void)
{
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
#if defined(__x86_64__)
    unsigned int mxcsr = _mm_getcsr();
    mxcsr &= ~(_MM_FLUSH_ZERO_ON | 0x0040u /* DAZ */);
    _mm_setcsr(mxcsr);
#endif
  }
}

/* the export interpolator preference (see ref_shim/common/conf.h) */
static const char *ref_interpolator = "mitchell";
void ref_set_interpolator(const char *name) { ref_interpolator = name; }
const char *dt_conf_get_string_const(const char *name)
{
  (void)name;
  return ref_interpolator;
}
