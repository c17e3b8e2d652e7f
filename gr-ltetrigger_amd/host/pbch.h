// PBCH / MIB decode at 6 PRB (1.92 Msps), host side.
//
// Replaces what lib/mib_impl.cc:148-170 asks of srsLTE (srslte_ue_mib_decode on the first
// subframe of an emitted half-frame, srslte_pbch_mib_unpack).  It runs once per newly tracked
// cell on a 1920-sample subframe, so it stays on the CPU like the reference's.
// Written from 3GPP TS 36.211 6.6 / 6.10.1 / 7.2 and TS 36.212 5.1.1 / 5.1.3.1 / 5.1.4.2 / 5.3.1.
#pragma once
#include <complex>
#include <cstdint>

namespace lt {
namespace pbch {

struct Mib {
  int nof_prb = 0;            // 6, 15, 25, 50, 75, 100
  int nof_ports = 0;          // 1, 2, 4 (from the CRC mask)
  int phich_length_ext = 0;   // 0 normal, 1 extended
  int phich_resources = 0;    // index into {1/6, 1/2, 1, 2}
  int sfn = 0;                // 8 MSBs of the SFN, shifted left by 2 (srslte_pbch_mib_unpack)
  int frame_idx = 0;          // which quarter of the 40 ms scrambling period matched
  uint8_t payload[24] = {0};  // BCH payload bits
};

// subframe: 1920 samples starting at the slot-0 boundary of the candidate subframe 0.
// nof_ports_hint: 0 = try 1, 2 and 4 (d_cell.nof_ports = 0 "detects nof_ports", lib/mib_impl.cc:69);
// otherwise only that count.  Returns true when a CRC matches.
bool decode_mib(const std::complex<float> *subframe, int cell_id, bool cp_ext, int nof_ports_hint, Mib *out);

} // namespace pbch
} // namespace lt
