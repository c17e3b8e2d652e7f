// gr::ltetrigger::sss on the gfx950 library: per half-frame CP detection, SSS m0/m1 detection and
// the cell_id / cp_type stream tags (reference flow: lib/sss_impl.cc:83-156).
#include "sss_impl.h"

#include <cstring>
#include <stdexcept>
#include <string>

namespace gr {
namespace ltetrigger {

const pmt::pmt_t sss_impl::cell_id_tag_key = pmt::intern("cell_id");
const pmt::pmt_t sss_impl::cp_type_tag_key = pmt::intern("cp_type");
const pmt::pmt_t sss_impl::tracking_lost_tag_key = pmt::intern("tracking_lost");

sss::sptr sss::make(int N_id_2) { return gnuradio::get_initial_sptr(new sss_impl(N_id_2, 0)); }

sss_impl::sss_impl(int N_id_2, int device)
    : gr::sync_block("sss", gr::io_signature::make(1, 1, sizeof(lt_cf32)), gr::io_signature::make(1, 1, sizeof(lt_cf32))),
      d_N_id_2(N_id_2)
{
  set_tag_propagation_policy(TPP_ALL_TO_ALL); // "tracking_lost" rides through to mib
  if (lt_sync_init(&d_sync, device, half_frame_length, max_offset, symbol_sz))
    throw std::runtime_error(std::string("Error initializing SSS SYNC: ") + lt_last_error());
  if (lt_sync_set_N_id_2(d_sync, N_id_2)) {
    lt_sync_free(d_sync);
    throw std::runtime_error("Error initializing SSS SYNC N_id_2");
  }
  set_output_multiple(half_frame_length);
}

sss_impl::~sss_impl() { lt_sync_free(d_sync); }

int sss_impl::work(int, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
{
  const lt_cf32 *in = static_cast<const lt_cf32 *>(input_items[0]);
  lt_cf32 *out = static_cast<lt_cf32 *>(output_items[0]);
  // exactly one half-frame per call, whatever the scheduler offers
  std::memcpy(out, in, sizeof(lt_cf32) * half_frame_length);

  get_tags_in_window(d_lost_tags, 0, 0, 1, tracking_lost_tag_key);
  if (!d_lost_tags.empty()) { // upstream lost the cell: forget the CP statistics, tag nothing
    lt_sync_reset(d_sync);
    d_lost_tags.clear();
    return half_frame_length;
  }

  const int cp = lt_sync_detect_cp(d_sync, in, slot_length);
  if (cp < 0) throw std::runtime_error(std::string("lt_sync_detect_cp: ") + lt_last_error());
  lt_sync_set_cp(d_sync, cp);

  // SSS is the symbol before the PSS, which ends the first slot
  const int sss_idx = slot_length - 2 * symbol_sz - lt_sync_get_cp_len(d_sync);
  uint32_t m0 = 0, m1 = 0;
  float m0_value = 0.f, m1_value = 0.f;
  if (lt_sss_m0m1_partial(d_sync, in + sss_idx, &m0, &m0_value, &m1, &m1_value))
    throw std::runtime_error(std::string("lt_sss_m0m1_partial: ") + lt_last_error());

  if (lt_sss_N_id_1(d_sync, m0, m1) < 0) return half_frame_length; // untagged: mib ignores it

  const int cell_id = lt_sync_get_cell_id(d_sync);
  add_item_tag(0, nitems_written(0), cell_id_tag_key, pmt::from_long(cell_id));
  add_item_tag(0, nitems_written(0), cp_type_tag_key, cp == LT_CP_NORM ? PMT_T : PMT_F);
  return half_frame_length;
}

} // namespace ltetrigger
} // namespace gr
