// gr::filter::rational_resampler_ccc(interpolation = 1, decimation = N) with GNU Radio's default
// taps, computed by the gfx950 decimator (lt_decim_*).  Used in front of the trigger exactly as
// examples/cell_search_file.py:56-57 and python/qa_downlink_trigger_c.py:109-112 do.
#pragma once
#include <memory>
#include <vector>

#include "../../include/ltetrigger_hip.h"
#include "runtime.h"

namespace gr {
namespace filter {

class rational_resampler_ccc : public gr::block
{
public:
  typedef std::shared_ptr<rational_resampler_ccc> sptr;
  static sptr make(unsigned interpolation, unsigned decimation, int device = 0);
  ~rational_resampler_ccc() override;
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override;
  void forecast(int noutput_items, gr_vector_int &ninput_items_required) override;

private:
  rational_resampler_ccc(unsigned decimation, int device);
  lt_decim *d_decim = nullptr;
  int d_D;
  int d_hist; // multiple of D, >= ntaps - 1
  std::vector<lt_cf32> d_tmp;
};

} // namespace filter
} // namespace gr
