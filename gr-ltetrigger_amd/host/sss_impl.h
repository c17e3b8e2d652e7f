// Private implementation of gr::ltetrigger::sss (reference state: lib/sss_impl.h:39-53).
#pragma once
#include <vector>

#include "../../include/ltetrigger_hip.h"
#include "ltetrigger/sss.h"

namespace gr {
namespace ltetrigger {

class sss_impl : public sss
{
public:
  sss_impl(int N_id_2, int device);
  ~sss_impl() override;
  int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) override;

  static const pmt::pmt_t cell_id_tag_key, cp_type_tag_key, tracking_lost_tag_key;

private:
  static constexpr int slot_length = LT_SLOT_LEN;
  static constexpr int half_frame_length = LT_HALF_FRAME_LEN;
  static constexpr int symbol_sz = LT_SYMBOL_SZ;
  static constexpr int max_offset = 100; // as arbitrary here as in lib/sss_impl.h:47

  lt_sync *d_sync = nullptr;
  int d_N_id_2;
  std::vector<tag_t> d_lost_tags;
};

} // namespace ltetrigger
} // namespace gr
