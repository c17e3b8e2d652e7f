// gr::ltetrigger::mib: consumes the tagged half-frames of sss, decodes PBCH/MIB on the first
// subframe of a newly identified cell, and publishes "track" / "drop" dictionaries
// (reference flow: lib/mib_impl.cc:97-183, dictionary lib/mib_impl.cc:185-251).
#include "mib_impl.h"

#include <complex>
#include <ctime>
#include <string>

namespace gr {
namespace ltetrigger {

const pmt::pmt_t mib_impl::cell_id_tag_key = pmt::intern("cell_id");
const pmt::pmt_t mib_impl::cp_type_tag_key = pmt::intern("cp_type");
const pmt::pmt_t mib_impl::tracking_lost_tag_key = pmt::intern("tracking_lost");
const pmt::pmt_t mib_impl::drop_port_id = pmt::intern("drop");
const pmt::pmt_t mib_impl::track_port_id = pmt::intern("track");

mib::sptr mib::make(bool exit_on_success) { return gnuradio::get_initial_sptr(new mib_impl(exit_on_success)); }

mib_impl::mib_impl(bool exit_on_success)
    : gr::block("mib", gr::io_signature::make(1, 1, sizeof(lt_cf32)), gr::io_signature::make(1, 1, sizeof(lt_cf32))),
      d_exit_on_success(exit_on_success)
{
  set_tag_propagation_policy(TPP_DONT);
  set_output_multiple(half_frame_length);
  message_port_register_out(track_port_id);
  message_port_register_out(drop_port_id);
}

int mib_impl::general_work(int noutput_items, gr_vector_int &, gr_vector_const_void_star &input_items,
                           gr_vector_void_star &)
{
  const std::complex<float> *in = static_cast<const std::complex<float> *>(input_items[0]);

  get_tags_in_window(d_lost_tags, 0, 0, 1, tracking_lost_tag_key);
  if (!d_lost_tags.empty()) { // the chain lost its cell: withdraw it and start over
    if (d_cell_published) message_port_pub(drop_port_id, d_current_tracking_cell);
    d_cell_published = false;
    d_current_tracking_cell = PMT_NIL;
    d_lost_tags.clear();
    consume_each(half_frame_length);
    return 0;
  }
  if (d_cell_published) { // already reported; nothing more to learn until it is lost
    consume_each(half_frame_length);
    return 0;
  }

  get_tags_in_window(d_cell_id_tags, 0, 0, 1, cell_id_tag_key);
  get_tags_in_window(d_cp_type_tags, 0, 0, 1, cp_type_tag_key);
  if (d_cell_id_tags.size() != 1 || d_cp_type_tags.size() != 1) { // junk through SSS
    consume_each(half_frame_length);
    return 0;
  }
  const unsigned cell_id = (unsigned)pmt::to_long(d_cell_id_tags[0].value);
  const bool cp_ext = !pmt::eq(d_cp_type_tags[0].value, PMT_T);
  d_cell_id_tags.clear();
  d_cp_type_tags.clear();

  if (cell_id != d_cell.id || cp_ext != d_cell.cp_ext) { // a different cell: new decoder context
    d_cell.id = cell_id;
    d_cell.cp_ext = cp_ext;
  }

  // single-frame decode only: the reference resets the PBCH accumulator before every attempt
  lt::pbch::Mib m;
  if (lt::pbch::decode_mib(in, (int)d_cell.id, d_cell.cp_ext, d_cell.nof_ports, &m)) {
    d_cell.nof_ports = m.nof_ports;
    d_cell.nof_prb = m.nof_prb;
    d_cell.phich_length_ext = m.phich_length_ext;
    d_cell.phich_resources = m.phich_resources;
    d_sfn_offset = m.sfn; // srslte_pbch_mib_unpack overwrites the offset with SFN<<2 (lib/mib_impl.cc:168-170)
    d_current_tracking_cell = pack_cell(d_cell, d_sfn_offset);
    message_port_pub(track_port_id, d_current_tracking_cell);
    d_cell_published = true;
    if (d_exit_on_success) return WORK_DONE;
  }
  consume_each(noutput_items);
  return noutput_items;
}

pmt::pmt_t mib_impl::pack_cell(const cell_t &cell, int sfn) const
{
  static const char *const phich_r[4] = {"1/6", "1/2", "1", "2"};
  pmt::pmt_t d = pmt::make_dict();
  d = pmt::dict_add(d, pmt::intern("cell_id"), pmt::from_long(cell.id));
  d = pmt::dict_add(d, pmt::intern("nof_tx_ports"), pmt::from_long(cell.nof_ports));
  d = pmt::dict_add(d, pmt::intern("cp_len"), pmt::intern(cell.cp_ext ? "Extended" : "Normal"));
  d = pmt::dict_add(d, pmt::intern("nof_prb"), pmt::from_long(cell.nof_prb));
  d = pmt::dict_add(d, pmt::intern("phich_len"), pmt::intern(cell.phich_length_ext ? "Extended" : "Normal"));
  d = pmt::dict_add(d, pmt::intern("nof_phich_resources"), pmt::intern(phich_r[cell.phich_resources & 3]));
  d = pmt::dict_add(d, pmt::intern("sfn_offset"), pmt::from_long(sfn));
  d = pmt::dict_add(d, pmt::intern("tracking_start_time"), pmt::from_long((long)std::time(nullptr)));
  return d;
}

} // namespace ltetrigger
} // namespace gr
