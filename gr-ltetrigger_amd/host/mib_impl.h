// Private implementation of gr::ltetrigger::mib (reference state: lib/mib_impl.h:39-82).
#pragma once
#include <vector>

#include "../../include/ltetrigger_hip.h"
#include "ltetrigger/mib.h"
#include "pbch.h"

namespace gr {
namespace ltetrigger {

class mib_impl : public mib
{
public:
  explicit mib_impl(bool exit_on_success);
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override;

  static const pmt::pmt_t cell_id_tag_key, cp_type_tag_key, tracking_lost_tag_key;
  static const pmt::pmt_t drop_port_id, track_port_id;

private:
  static constexpr int half_frame_length = LT_HALF_FRAME_LEN;
  struct cell_t {
    unsigned id = 0;
    bool cp_ext = false;
    int nof_ports = 0; // 0: detect (lib/mib_impl.cc:69)
    int nof_prb = 6;
    int phich_length_ext = 0;
    int phich_resources = 0;
  };
  pmt::pmt_t pack_cell(const cell_t &cell, int sfn) const;

  pmt::pmt_t d_current_tracking_cell = PMT_NIL;
  bool d_cell_published = false;
  bool d_exit_on_success;
  cell_t d_cell;
  int d_sfn_offset = 0;
  std::vector<tag_t> d_lost_tags, d_cell_id_tags, d_cp_type_tags;
};

} // namespace ltetrigger
} // namespace gr
