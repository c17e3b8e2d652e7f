#include "capi.h"

#include <cstdio>
#include <cstring>
#include <exception>
#include <sstream>
#include <string>
#include <vector>

#include "cellstore_impl.h"
#include "mib_impl.h"
#include "pbch.h"
#include "pss_impl.h"
#include "resampler.h"
#include "runtime.h"
#include "sss_impl.h"

namespace {

thread_local std::string g_err;

// sink that records everything (items and tags) for inspection from Python
class probe : public gr::sync_block
{
public:
  probe() : gr::sync_block("probe", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(0, 0, 0)) {}
  int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &) override
  {
    const gr_complex *p = static_cast<const gr_complex *>(in[0]);
    std::vector<gr::tag_t> t;
    get_tags_in_range(t, 0, nitems_read(0), nitems_read(0) + (uint64_t)n);
    tags.insert(tags.end(), t.begin(), t.end());
    items.insert(items.end(), p, p + n);
    return n;
  }
  std::vector<gr_complex> items;
  std::vector<gr::tag_t> tags;
};

int copy_out(const std::string &s, char *buf, int len)
{
  if ((int)s.size() + 1 > len) {
    g_err = "buffer too small";
    return -(int)s.size() - 1;
  }
  std::memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

} // namespace

struct ltr_tb {
  gr::top_block tb;
  std::vector<gr::basic_block_sptr> blocks;
  int add(gr::basic_block_sptr b)
  {
    blocks.push_back(b);
    return (int)blocks.size() - 1;
  }
  template <class T> std::shared_ptr<T> get(int id)
  {
    if (id < 0 || id >= (int)blocks.size()) throw std::runtime_error("bad block id");
    auto p = std::dynamic_pointer_cast<T>(blocks[(size_t)id]);
    if (!p) throw std::runtime_error("block " + std::to_string(id) + " has the wrong type");
    return p;
  }
};

#define LTR_TRY(body, onerr)                \
  try {                                     \
    body                                    \
  } catch (const std::exception &e) {       \
    g_err = e.what();                       \
    return onerr;                           \
  }

extern "C" {

ltr_tb *ltr_tb_create(void) { return new ltr_tb; }
void ltr_tb_free(ltr_tb *tb) { delete tb; }
const char *ltr_last_error(void) { return g_err.c_str(); }

int ltr_add_vector_source(ltr_tb *tb, const lt_cf32 *data, int64_t n, int repeat)
{
  LTR_TRY(const gr_complex *p = reinterpret_cast<const gr_complex *>(data);
          return tb->add(gr::blocks::vector_source_c::make(std::vector<gr_complex>(p, p + n), repeat != 0));, -1)
}
int ltr_add_file_source(ltr_tb *tb, const char *path, int repeat)
{
  LTR_TRY(return tb->add(gr::blocks::file_source(path, repeat != 0));, -1)
}
int ltr_add_head(ltr_tb *tb, int64_t nitems) { LTR_TRY(return tb->add(gr::blocks::head::make((uint64_t)nitems));, -1) }
int ltr_add_resampler(ltr_tb *tb, int decimation)
{
  LTR_TRY(return tb->add(gr::filter::rational_resampler_ccc::make(1, (unsigned)decimation));, -1)
}
int ltr_add_pss(ltr_tb *tb, int N_id_2, float thr, int track_after, int track_every)
{
  LTR_TRY(return tb->add(gr::ltetrigger::pss::make(N_id_2, thr, track_after, track_every));, -1)
}
int ltr_add_sss(ltr_tb *tb, int N_id_2) { LTR_TRY(return tb->add(gr::ltetrigger::sss::make(N_id_2));, -1) }
int ltr_add_mib(ltr_tb *tb, int exit_on_success) { LTR_TRY(return tb->add(gr::ltetrigger::mib::make(exit_on_success != 0));, -1) }
int ltr_add_cellstore(ltr_tb *tb) { LTR_TRY(return tb->add(gr::ltetrigger::cellstore::make());, -1) }
int ltr_add_message_debug(ltr_tb *tb) { LTR_TRY(return tb->add(gr::blocks::message_debug::make());, -1) }
int ltr_add_null_sink(ltr_tb *tb) { LTR_TRY(return tb->add(gr::blocks::null_sink::make());, -1) }
int ltr_add_probe(ltr_tb *tb) { LTR_TRY(return tb->add(std::make_shared<probe>());, -1) }

int ltr_connect(ltr_tb *tb, int src, int dst)
{
  LTR_TRY(tb->tb.connect(tb->get<gr::block>(src), tb->get<gr::block>(dst)); return 0;, -1)
}
int ltr_msg_connect(ltr_tb *tb, int src, const char *sp, int dst, const char *dp)
{
  LTR_TRY(tb->tb.msg_connect(tb->get<gr::basic_block>(src), sp, tb->get<gr::basic_block>(dst), dp); return 0;, -1)
}
int ltr_run(ltr_tb *tb) { LTR_TRY(tb->tb.run(); return 0;, -1) }
int ltr_run_timeout(ltr_tb *tb, double s) { LTR_TRY(tb->tb.run(s); return 0;, -1) }
int ltr_run_threaded(ltr_tb *tb, double s) { LTR_TRY(tb->tb.run_threaded(s); return 0;, -1) }

float ltr_pss_max_psr(ltr_tb *tb, int id) { LTR_TRY(return tb->get<gr::ltetrigger::pss>(id)->max_psr();, -1.f) }
float ltr_pss_mean_psr(ltr_tb *tb, int id) { LTR_TRY(return tb->get<gr::ltetrigger::pss>(id)->mean_psr();, -1.f) }
float ltr_pss_mean_cfo(ltr_tb *tb, int id) { LTR_TRY(return tb->get<gr::ltetrigger::pss>(id)->mean_cfo();, -1.f) }
float ltr_pss_psr_threshold(ltr_tb *tb, int id) { LTR_TRY(return tb->get<gr::ltetrigger::pss>(id)->psr_threshold();, -1.f) }
float ltr_pss_tracking_score(ltr_tb *tb, int id) { LTR_TRY(return tb->get<gr::ltetrigger::pss>(id)->tracking_score();, -1.f) }
int ltr_pss_set_psr_threshold(ltr_tb *tb, int id, float t)
{
  LTR_TRY(tb->get<gr::ltetrigger::pss>(id)->set_psr_threshold(t); return 0;, -1)
}

int ltr_msgdebug_count(ltr_tb *tb, int id) { LTR_TRY(return tb->get<gr::blocks::message_debug>(id)->num_messages();, -1) }
int ltr_msgdebug_json(ltr_tb *tb, int id, int idx, char *buf, int len)
{
  LTR_TRY(return copy_out(pmt::to_json(tb->get<gr::blocks::message_debug>(id)->get_message(idx)), buf, len);, -1)
}

int ltr_cellstore_tracking(ltr_tb *tb, int id) { LTR_TRY(return tb->get<gr::ltetrigger::cellstore>(id)->tracking() ? 1 : 0;, -1) }
int ltr_cellstore_cells_json(ltr_tb *tb, int id, char *buf, int len)
{
  LTR_TRY(std::ostringstream o; o << '['; bool first = true;
          for (auto &c : tb->get<gr::ltetrigger::cellstore>(id)->cells()) {
            if (!first) o << ", ";
            first = false;
            o << pmt::to_json(c);
          } o << ']';
          return copy_out(o.str(), buf, len);, -1)
}
int ltr_cellstore_latest_json(ltr_tb *tb, int id, char *buf, int len)
{
  LTR_TRY(return copy_out(pmt::to_json(tb->get<gr::ltetrigger::cellstore>(id)->latest_cell()), buf, len);, -1)
}

int64_t ltr_probe_nitems(ltr_tb *tb, int id) { LTR_TRY(return (int64_t)tb->get<probe>(id)->items.size();, -1) }
int64_t ltr_probe_items(ltr_tb *tb, int id, lt_cf32 *out, int64_t cap)
{
  LTR_TRY(auto p = tb->get<probe>(id); int64_t n = (int64_t)p->items.size(); if (n > cap) n = cap;
          std::memcpy(out, p->items.data(), sizeof(lt_cf32) * (size_t)n); return n;, -1)
}
int ltr_probe_tags_json(ltr_tb *tb, int id, char *buf, int len)
{
  LTR_TRY(std::ostringstream o; o << '['; bool first = true;
          for (auto &t : tb->get<probe>(id)->tags) {
            if (!first) o << ", ";
            first = false;
            o << "{\"offset\": " << t.offset << ", \"key\": \"" << pmt::symbol_to_string(t.key)
              << "\", \"value\": " << pmt::to_json(t.value) << '}';
          } o << ']';
          return copy_out(o.str(), buf, len);, -1)
}

int ltr_decode_mib(const lt_cf32 *sf, int cell_id, int cp_ext, int hint, ltr_mib *out)
{
  LTR_TRY(lt::pbch::Mib m;
          if (!lt::pbch::decode_mib(reinterpret_cast<const std::complex<float> *>(sf), cell_id, cp_ext != 0, hint, &m)) return 0;
          out->nof_prb = m.nof_prb; out->nof_ports = m.nof_ports; out->phich_length_ext = m.phich_length_ext;
          out->phich_resources = m.phich_resources; out->sfn = m.sfn; out->frame_idx = m.frame_idx;
          return 1;, -1)
}

} // extern "C"
