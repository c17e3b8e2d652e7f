// In-repo mini runtime with GNU Radio 3.7's block interface, used because GNU Radio is absent from
// both containers: the block sources in this directory are written against the
// GNU Radio API (gr::block / gr::sync_block, general_work / work, set_history,
// set_output_multiple, stream tags, message ports) and this header supplies just enough of it to
// run them: complex-float streams only, one input and one output per block.  Two schedulers: run() steps the blocks in
// flowgraph order on the caller's thread (the call sequence of every block is then a function of the input alone: what
// the parity tests use); run_threaded() gives every block a thread of its own, as GNU Radio's thread-per-block
// scheduler does -- work() of different blocks runs concurrently, getters are polled from foreign threads
// (lib/pss_impl.h:95-100), message handlers run under the receiving block's message lock.
// Semantics follow SURVEY Appendix A.7.
#pragma once
#include <atomic>
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "pmt_lite.h"

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace gnuradio {
template <class T> std::shared_ptr<T> get_initial_sptr(T *p) { return std::shared_ptr<T>(p); }
} // namespace gnuradio

namespace gr {

struct tag_t {
  uint64_t offset = 0;
  pmt::pmt_t key, value, srcid;
};

struct io_signature {
  typedef std::shared_ptr<io_signature> sptr;
  int min_streams, max_streams, item_size;
  static sptr make(int mn, int mx, int sz) { return sptr(new io_signature{mn, mx, sz}); }
};

class top_block;

class basic_block : public std::enable_shared_from_this<basic_block>
{
public:
  typedef std::function<void(pmt::pmt_t)> msg_handler_t;
  explicit basic_block(const std::string &name) : d_name(name) {}
  virtual ~basic_block() {}
  const std::string &name() const { return d_name; }
  void message_port_register_out(pmt::pmt_t port) { d_out_ports[port.get()]; }
  void message_port_register_in(pmt::pmt_t port) { d_in_ports[port.get()]; }
  void set_msg_handler(pmt::pmt_t port, msg_handler_t h) { d_in_ports[port.get()] = h; }
  void message_port_pub(pmt::pmt_t port, pmt::pmt_t msg);
  void post(pmt::pmt_t port, pmt::pmt_t msg);

private:
  friend class top_block;
  std::string d_name;
  std::map<const pmt::pmt_base *, std::vector<std::pair<basic_block *, pmt::pmt_t>>> d_out_ports;
  std::map<const pmt::pmt_base *, msg_handler_t> d_in_ports;
  std::mutex d_msg_mu; // a block's message handlers run one at a time, whichever thread posts (GNU Radio: on the block's own thread)
};
typedef std::shared_ptr<basic_block> basic_block_sptr;

class block : public basic_block
{
public:
  enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
  enum tag_propagation_policy_t { TPP_DONT = 0, TPP_ALL_TO_ALL = 1, TPP_ONE_TO_ONE = 2 };

  block(const std::string &name, io_signature::sptr in, io_signature::sptr out);

  virtual int general_work(int noutput_items, gr_vector_int &ninput_items,
                           gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
  virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required);

  unsigned history() const { return d_history; }
  void set_history(unsigned h);
  void set_output_multiple(int m) { d_output_multiple = m; }
  int output_multiple() const { return d_output_multiple; }
  void consume_each(int n) { d_consumed += n; }
  void consume(int, int n) { d_consumed += n; }
  uint64_t nitems_read(unsigned) const { return (uint64_t)d_nread; }
  uint64_t nitems_written(unsigned) const { return d_nwritten; }
  void add_item_tag(unsigned which_output, uint64_t abs_offset, const pmt::pmt_t &key,
                    const pmt::pmt_t &value, const pmt::pmt_t &srcid = PMT_F);
  void get_tags_in_window(std::vector<tag_t> &v, unsigned which_input, uint64_t rel_start,
                          uint64_t rel_end, const pmt::pmt_t &key);
  void get_tags_in_range(std::vector<tag_t> &v, unsigned which_input, uint64_t abs_start, uint64_t abs_end);
  void set_tag_propagation_policy(tag_propagation_policy_t p) { d_tpp = p; }
  tag_propagation_policy_t tag_propagation_policy() const { return d_tpp; }
  bool done() const { return d_done.load(std::memory_order_acquire); }
  // true while the scheduler calls general_work with less input than forecast() asked for
  // because the upstream has finished (end of a finite capture)
  bool draining() const { return d_draining; }

protected:
  bool is_source() const { return d_in_sig->max_streams == 0; }

private:
  friend class top_block;
  bool step();
  void accept(const gr_complex *items, size_t n, const std::vector<tag_t> &tags);
  void take_inbox();
  size_t backlog() const { return d_backlog.load(std::memory_order_relaxed); }
  void thread_main(top_block *tb);

  io_signature::sptr d_in_sig, d_out_sig;
  unsigned d_history = 1;
  int d_output_multiple = 1;
  tag_propagation_policy_t d_tpp = TPP_ALL_TO_ALL;
  // input side.  d_in / d_in_tags belong to the thread that steps the block; what the upstream hands over lands in the inbox
  // (under d_mu, from the upstream's thread) and is moved across at the top of a step.
  std::vector<gr_complex> d_in; // d_in[0] is absolute item d_in_abs0 (negative while history is zeros)
  int64_t d_in_abs0 = 0;
  int64_t d_nread = 0;
  std::vector<tag_t> d_in_tags;
  bool d_upstream_done = false;
  std::mutex d_mu;
  std::vector<gr_complex> d_inbox;
  std::vector<tag_t> d_inbox_tags;
  std::atomic<size_t> d_backlog{0}; // items handed over and not yet consumed
  // output side
  uint64_t d_nwritten = 0;
  std::vector<tag_t> d_new_tags;
  std::vector<gr_complex> d_out;
  int d_consumed = 0;
  std::atomic<bool> d_done{false};
  bool d_draining = false;
  block *d_upstream = nullptr;
  std::vector<block *> d_downstream;
};
typedef std::shared_ptr<block> block_sptr;

class sync_block : public block
{
public:
  sync_block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : block(name, in, out) {}
  virtual int work(int noutput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) = 0;
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override;
};

class top_block
{
public:
  void connect(block_sptr src, block_sptr dst);
  void msg_connect(basic_block_sptr src, const std::string &src_port, basic_block_sptr dst,
                   const std::string &dst_port);
  // run to completion, or (max_seconds >= 0) until that much wall time has passed -- tb.start();
  // sleep; tb.stop(); tb.wait() of examples/cell_search_file.py:101-111
  void run(double max_seconds = -1.0);
  // the same flowgraph with one thread per block (GNU Radio's default scheduler): returns when every block is done, or
  // after max_seconds.  An exception thrown by a block's work ends the run and is rethrown here.
  void run_threaded(double max_seconds = -1.0);

private:
  friend class block;
  std::vector<block_sptr> d_blocks; // in connection order; sources first by construction
  std::vector<basic_block_sptr> d_keepalive;
  void add(block_sptr b);
  // run_threaded: one condition variable for the whole flowgraph -- a block that cannot step waits on it, every step that
  // moved items (or finished a block) signals it
  std::mutex d_wake_mu;
  std::condition_variable d_wake;
  uint64_t d_wake_gen = 0;
  std::atomic<bool> d_stop{false};
  void wake_all();
};

// ---- the handful of stock blocks the reference's flowgraphs use
// (python/qa_downlink_trigger_c.py:70-81, examples/cell_search_file.py:41-62)
namespace blocks {

class vector_source_c : public sync_block
{
public:
  typedef std::shared_ptr<vector_source_c> sptr;
  static sptr make(const std::vector<gr_complex> &data, bool repeat);
  int work(int, gr_vector_const_void_star &, gr_vector_void_star &) override;

private:
  vector_source_c(const std::vector<gr_complex> &d, bool r);
  std::vector<gr_complex> d_data;
  bool d_repeat;
  size_t d_pos = 0;
};

// blocks.file_source(gr.sizeof_gr_complex, filename, repeat)
vector_source_c::sptr file_source(const std::string &filename, bool repeat);

class head : public sync_block
{
public:
  typedef std::shared_ptr<head> sptr;
  static sptr make(uint64_t nitems) { return sptr(new head(nitems)); }
  int work(int, gr_vector_const_void_star &, gr_vector_void_star &) override;

private:
  explicit head(uint64_t n);
  uint64_t d_nitems, d_count = 0;
};

class null_sink : public sync_block // stands in for blocks.tag_debug with display off
{
public:
  typedef std::shared_ptr<null_sink> sptr;
  static sptr make() { return sptr(new null_sink); }
  int work(int n, gr_vector_const_void_star &, gr_vector_void_star &) override { return n; }

private:
  null_sink();
};

class message_debug : public basic_block
{
public:
  typedef std::shared_ptr<message_debug> sptr;
  static sptr make() { return sptr(new message_debug); }
  int num_messages() const { std::lock_guard<std::mutex> l(d_mu); return (int)d_msgs.size(); }
  pmt::pmt_t get_message(int i) const { std::lock_guard<std::mutex> l(d_mu); return d_msgs.at(i); }

private:
  message_debug();
  mutable std::mutex d_mu; // (read from the application's thread while the flowgraph runs)
  std::vector<pmt::pmt_t> d_msgs;
};

} // namespace blocks
} // namespace gr
