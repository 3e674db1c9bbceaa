// pipe.cpp -- export-pipe executor on the dt_hip_* runtime: the device-resident part of
// dt_dev_pixelpipe_process_rec() (src/develop/pixelpipe_hb.c:881-1282) + pixelpipe_process_on_GPU()
// (src/develop/pixelpipe_gpu.c:191-744) for an export, where no module output is re-used.
//
// The reference walks the node list recursively from the last node, acquires a cacheline per module
// output and calls process_cl(); here the list is walked forward, outputs come from the runtime's
// pool and go back to it as soon as the consumer is enqueued (stream-ordered, so safe), and runs
// of pointwise modules are planned into fused launches (pipe_fused.hip).
#include "hip_common.h"
#include "pipe_fused.h"

#include <string>
#include <vector>

using namespace ansel;

namespace
{

enum op_t
{
  OP_RAWPREPARE,
  OP_TEMPERATURE,
  OP_HIGHLIGHTS,
  OP_DEMOSAIC,
  OP_EXPOSURE,
  OP_COLORIN,
  OP_CHANNELMIXERRGB,
  OP_FILMICRGB,
  OP_COLOROUT,
  OP_EXPORT_U16,
  OP_UNKNOWN
};

struct op_info_t
{
  const char *name;
  size_t data_size;
  int bpp_out; // bytes per pixel of the module output
};

const op_info_t k_ops[] = {
  { "rawprepare", sizeof(dt_hip_rawprepare_data_t), 4 },
  { "temperature", sizeof(dt_hip_temperature_data_t), 0 /* = input */ },
  { "highlights", sizeof(dt_hip_highlights_data_t), 0 },
  { "demosaic", sizeof(dt_hip_demosaic_data_t), 16 },
  { "exposure", sizeof(dt_hip_exposure_data_t), 0 },
  { "colorin", sizeof(dt_hip_conversion_t), 16 },
  { "channelmixerrgb", sizeof(dt_hip_channelmixerrgb_data_t), 16 },
  { "filmicrgb", sizeof(dt_hip_filmicrgb_data_t), 16 },
  { "colorout", sizeof(dt_hip_conversion_t), 16 },
  { "export_u16", 0, 8 },
};

struct node_t
{
  op_t op;
  dt_hip_piece_t piece;
  std::vector<unsigned char> data;
  template <typename T> const T *as() const { return reinterpret_cast<const T *>(data.data()); }
};

struct group_t
{
  enum kind_t { SINGLE, RAW, RGB } kind;
  int first, count; // node range
  raw_group_t raw;
  rgb_group_t rgb;
};

size_t out_bytes(const node_t &n)
{
  const size_t px = (size_t)n.piece.roi_out.width * n.piece.roi_out.height;
  switch(n.op)
  {
    case OP_RAWPREPARE: return px * 4;
    case OP_TEMPERATURE:
    case OP_HIGHLIGHTS:
    case OP_EXPOSURE: return px * 4 * n.piece.channels;
    case OP_EXPORT_U16: return px * 8;
    default: return px * 16;
  }
}

int run_single(int devid, const node_t &n, dt_hip_mem_t in, dt_hip_mem_t out)
{
  switch(n.op)
  {
    case OP_RAWPREPARE: return dt_hip_iop_rawprepare_process(devid, &n.piece, n.as<dt_hip_rawprepare_data_t>(), in, out);
    case OP_TEMPERATURE: return dt_hip_iop_temperature_process(devid, &n.piece, n.as<dt_hip_temperature_data_t>(), in, out);
    case OP_HIGHLIGHTS: return dt_hip_iop_highlights_process(devid, &n.piece, n.as<dt_hip_highlights_data_t>(), in, out);
    case OP_DEMOSAIC: return dt_hip_iop_demosaic_process(devid, &n.piece, n.as<dt_hip_demosaic_data_t>(), in, out);
    case OP_EXPOSURE: return dt_hip_iop_exposure_process(devid, &n.piece, n.as<dt_hip_exposure_data_t>(), in, out);
    case OP_COLORIN: return dt_hip_iop_colorin_process(devid, &n.piece, n.as<dt_hip_conversion_t>(), in, out);
    case OP_CHANNELMIXERRGB: return dt_hip_iop_channelmixerrgb_process(devid, &n.piece, n.as<dt_hip_channelmixerrgb_data_t>(), in, out);
    case OP_FILMICRGB: return dt_hip_iop_filmicrgb_process(devid, &n.piece, n.as<dt_hip_filmicrgb_data_t>(), in, out);
    case OP_COLOROUT: return dt_hip_iop_colorout_process(devid, &n.piece, n.as<dt_hip_conversion_t>(), in, out);
    case OP_EXPORT_U16: return dt_hip_export_convert_u16(devid, n.piece.roi_out.width, n.piece.roi_out.height, in, out);
    default: return DT_HIP_INVALID_ARG;
  }
}

} // namespace

struct dt_hip_pipe_t
{
  int devid;
  bool fusion;
  bool planned;
  std::vector<node_t> nodes;
  std::vector<group_t> groups;

  void plan()
  {
    groups.clear();
    const int n = (int)nodes.size();
    int i = 0;
    while(i < n)
    {
      group_t g;
      g.kind = group_t::SINGLE;
      g.first = i;
      g.count = 1;
      if(fusion && nodes[i].op == OP_RAWPREPARE)
      {
        raw_group_t r;
        memset(&r, 0, sizeof(r));
        r.rawprepare_piece = nodes[i].piece;
        r.rawprepare = *nodes[i].as<dt_hip_rawprepare_data_t>();
        int j = i + 1;
        if(j < n && nodes[j].op == OP_TEMPERATURE)
        {
          r.has_temperature = true;
          r.temperature_piece = nodes[j].piece;
          r.temperature = *nodes[j].as<dt_hip_temperature_data_t>();
          j++;
        }
        if(j < n && nodes[j].op == OP_HIGHLIGHTS)
        {
          r.has_highlights = true;
          r.highlights_piece = nodes[j].piece;
          r.highlights = *nodes[j].as<dt_hip_highlights_data_t>();
          j++;
        }
        if(j - i > 1 && raw_group_supported(r))
        {
          g.kind = group_t::RAW;
          g.count = j - i;
          g.raw = r;
        }
      }
      else if(fusion && nodes[i].op >= OP_EXPOSURE && nodes[i].op <= OP_COLOROUT && nodes[i].piece.channels == 4)
      {
        rgb_group_t r;
        memset(&r, 0, sizeof(r));
        r.width = nodes[i].piece.roi_out.width;
        r.height = nodes[i].piece.roi_out.height;
        // the fused kernel applies its stages in the reference's pipe order (exposure < colorin <
        // channelmixerrgb < filmicrgb < colorout, src/develop/iop_order.c); a run is fusable as long
        // as it walks that order
        int last_op = -1;
        int j = i;
        while(j < n && r.n_ops < 8)
        {
          const node_t &nd = nodes[j];
          if(nd.op < OP_EXPOSURE || nd.op > OP_COLOROUT || (int)nd.op <= last_op) break;
          if(nd.piece.roi_out.width != r.width || nd.piece.roi_out.height != r.height || nd.piece.channels != 4) break;
          if(nd.op == OP_FILMICRGB)
          {
            const int v = nd.as<dt_hip_filmicrgb_data_t>()->version;
            if(v < 3 || v > 9) break;
          }
          if(nd.op == OP_CHANNELMIXERRGB && nd.as<dt_hip_channelmixerrgb_data_t>()->adaptation > DT_HIP_ADAPTATION_RGB) break;
          last_op = (int)nd.op;
          switch(nd.op)
          {
            case OP_EXPOSURE: r.ops[r.n_ops++] = RGB_OP_EXPOSURE; r.exposure = *nd.as<dt_hip_exposure_data_t>(); break;
            case OP_COLORIN: r.ops[r.n_ops++] = RGB_OP_COLORIN; r.colorin = *nd.as<dt_hip_conversion_t>(); break;
            case OP_CHANNELMIXERRGB: r.ops[r.n_ops++] = RGB_OP_CHANNELMIXER; r.channelmixer = *nd.as<dt_hip_channelmixerrgb_data_t>(); break;
            case OP_FILMICRGB: r.ops[r.n_ops++] = RGB_OP_FILMIC; r.filmic = *nd.as<dt_hip_filmicrgb_data_t>(); break;
            default: r.ops[r.n_ops++] = RGB_OP_COLOROUT; r.colorout = *nd.as<dt_hip_conversion_t>(); break;
          }
          j++;
        }
        if(j < n && nodes[j].op == OP_EXPORT_U16 && nodes[j].piece.roi_out.width == r.width
           && nodes[j].piece.roi_out.height == r.height)
        {
          r.to_u16 = true;
          j++;
        }
        if(j - i > 1)
        {
          g.kind = group_t::RGB;
          g.count = j - i;
          g.rgb = r;
        }
      }
      groups.push_back(g);
      i += g.count;
    }
    planned = true;
  }
};

extern "C" {

dt_hip_pipe_t *dt_hip_pipe_new(int devid)
{
  if(!valid_device(devid)) return nullptr;
  dt_hip_pipe_t *p = new dt_hip_pipe_t;
  p->devid = devid;
  p->fusion = true;
  p->planned = false;
  return p;
}

void dt_hip_pipe_free(dt_hip_pipe_t *pipe) { delete pipe; }

int dt_hip_pipe_add_node(dt_hip_pipe_t *pipe, const char *op, const dt_hip_piece_t *piece, const void *data,
                         size_t data_size)
{
  if(!pipe || !op || !piece) return DT_HIP_INVALID_ARG;
  op_t o = OP_UNKNOWN;
  for(int k = 0; k < (int)OP_UNKNOWN; k++)
    if(!strcmp(op, k_ops[k].name)) o = (op_t)k;
  if(o == OP_UNKNOWN)
  {
    set_last_error("dt_hip_pipe_add_node: module '%s' has no device implementation", op);
    return DT_HIP_INVALID_ARG;
  }
  if(data_size != k_ops[o].data_size || (data_size && !data))
  {
    set_last_error("dt_hip_pipe_add_node: '%s' expects %zu bytes of data, got %zu", op, k_ops[o].data_size, data_size);
    return DT_HIP_INVALID_ARG;
  }
  node_t n;
  n.op = o;
  n.piece = *piece;
  if(data_size) n.data.assign((const unsigned char *)data, (const unsigned char *)data + data_size);
  pipe->nodes.push_back(n);
  pipe->planned = false;
  return DT_HIP_SUCCESS;
}

void dt_hip_pipe_set_fusion(dt_hip_pipe_t *pipe, int enabled)
{
  if(!pipe) return;
  pipe->fusion = enabled != 0;
  pipe->planned = false;
}

int dt_hip_pipe_num_groups(dt_hip_pipe_t *pipe)
{
  if(!pipe) return 0;
  if(!pipe->planned) pipe->plan();
  return (int)pipe->groups.size();
}

int dt_hip_pipe_process(dt_hip_pipe_t *pipe, dt_hip_mem_t dev_in, dt_hip_mem_t dev_out)
{
  if(!pipe || !dev_in || !dev_out) return DT_HIP_INVALID_ARG;
  if(pipe->nodes.empty()) return DT_HIP_SUCCESS;
  if(!pipe->planned) pipe->plan();
  const int devid = pipe->devid;
  dt_hip_mem_t cur = dev_in;
  bool cur_owned = false;
  const size_t ng = pipe->groups.size();
  for(size_t gi = 0; gi < ng; gi++)
  {
    const group_t &g = pipe->groups[gi];
    const node_t &last = pipe->nodes[g.first + g.count - 1];
    dt_hip_mem_t out = dev_out;
    bool out_owned = false;
    if(gi + 1 < ng)
    {
      out = dt_hip_alloc_device_buffer(devid, out_bytes(last));
      if(!out)
      {
        if(cur_owned) dt_hip_release_mem_object(cur);
        return DT_HIP_SYSMEM_ALLOCATION;
      }
      out_owned = true;
    }
    int err;
    if(g.kind == group_t::RAW)
      err = raw_group_launch(devid, g.raw, cur, out);
    else if(g.kind == group_t::RGB)
      err = rgb_group_launch(devid, g.rgb, cur, out);
    else
      err = run_single(devid, pipe->nodes[g.first], cur, out);
    if(cur_owned) dt_hip_release_mem_object(cur); // stream-ordered: re-used only by later launches
    if(err != DT_HIP_SUCCESS)
    {
      if(out_owned) dt_hip_release_mem_object(out);
      return err;
    }
    cur = out;
    cur_owned = out_owned;
  }
  return DT_HIP_SUCCESS;
}

} // extern "C"
