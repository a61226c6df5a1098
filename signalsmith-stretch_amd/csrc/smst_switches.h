// Cross-check switches of the engine, in ONE place.  Every one selects another FORM of the same computation (a kernel organisation that is
// kept as a cross-check of the default one: tests/test_parity_gpu.py compares the forms bit for bit and asserts through launch counters that
// both ran) or a diagnostic; none changes what is computed, none is needed by a user of the library.  They are read from the environment once,
// when a batch is created.  A build with -DSMST_NO_SWITCHES compiles the parsing out: every switch keeps its default (csrc/Makefile:
// `make EXTRA_FLAGS=-DSMST_NO_SWITCHES`), and the library contains no getenv() besides SMST_DEVICE (include/smst.h) and the workspace budget.
//
//   variable               default  meaning
//   SMST_NO_OVERLAP        0        1: the tiles of a call run on one HIP stream (no feed-forward / recurrence / synthesis overlap)
//   SMST_NO_FUSE           unset    set: record kernel + recurrence kernel with records through HBM (kPredictB + kChain) instead of kVocoder / kVocoderN
//   SMST_NO_SINGLE_HOP     unset    set: one-hop tiles through the wavefront kernels instead of kVocoderOne / ACROSS
//   SMST_NO_ACROSS         unset    set: one-hop tiles one chain per stream (kVocoderOne) instead of lanes across streams
//   SMST_CHECK_LAUNCHES    0        1: hipGetLastError() after every launch group of process(), not only at its end
//   SMST_NO_FEED_FUSION    0        1: pass A (the (P, E) rows) as its own kernel (kPredictA) instead of folded into the feed kernels; 2: formant tiles in two passes over
//                                   the spectra (round 5's form) even where every base frequency is given and one pass does (round 6)
//   SMST_NO_STAGE          unset    set: the fused kernel's producers gather from HBM where staging applies
//   SMST_NO_ALIGN          unset    set: staged producers with per-row windows and lag L + 1 instead of the line-aligned form
//   SMST_ALIGN_ALL         unset    set: the line-aligned producers for every geometry they are valid for (default: L = 4 only)
//   SMST_CONTINUOUS        unset    set: runs of plain stereo / mono tiles through ONE wavefront across the tiles (kVocoderCont) instead of tile by tile (kVocoder):
//                                   bit-identical, measured equal in speed (EXPERIMENTS.md 6.1), kept as a cross-check of the tile form
//   SMST_CONT_WRITER_WAVE  4        11: kVocoderCont's result writer on the SIMD that holds two producers instead of the recurrence wave's (measured: no difference, EXPERIMENTS.md 6.2)
//   SMST_NO_FAST_FFT       unset    set: the generic radix-4/2/3/5 ladder even where a register-blocked FFT exists
//   SMST_FFT_TABLES        full     lean: the smaller FFT tables (one more rounding per element: opt-in, see smst_engine.cpp)
//   SMST_FEED_SERIAL       unset    set: bin-by-bin feed recurrences (kFeedSerial) instead of the scan form
//   SMST_FFT_TEAMS         1        0: one frame per workgroup; 2: persistent teams even for tiles with few frames per team (tests)
//   SMST_SYNTH_EMIT        1        0: kSynthTeams + kEmit; 2: kSynthEmitTeams also for small tiles (tests)
//   SMST_CARRIED_EMIT      1        0: a call without hops emits through kEmit (which copies the carry) instead of kEmitCarried (cross-check: bit-identical)
//   SMST_VOCN_WIDE         1        0: kVocoderN's producer passes 16 rows x 4 steps (the first form) instead of 8 rows x 8 steps
//   SMST_DEBUG_MODE        0        timing experiments; only in builds with -DSMST_EXPERIMENTS
//   SMST_WORKSPACE_GIB     auto     tile workspace budget per workspace in GiB (a tuning knob, not a cross-check: always read)
//   SMST_SUB_STREAMS       auto     streams per sub-batch, if smaller than what the budget allows (the same kind of knob: sub-batches alternate between the
//                                   two tile workspaces, so the bulk kernels of one overlap the recurrence of the other)
#pragma once
#include <cstdlib>
#include <string>

namespace smst {

struct Switches {
	bool overlap = true, noFuse = false, noSingleHop = false, noAcross = false, checkLaunches = false;
	int noFeedFusion = 0, fftTeams = 1, synthEmit = 1, debugMode = 0, vocNWide = 1, vocWide = 1, vocNHalfLines = 2, carriedEmit = 1;
	bool noStage = false, noAlign = false, alignAll = false, noFastFft = false, fftLean = false, feedSerial = false, continuous = false;
	int contWriterWave = 4;
	double workspaceGiB = 0; // 0: automatic
	int subStreams = 0;      // 0: automatic

	static Switches fromEnvironment() {
		Switches s;
		if (const char *env = std::getenv("SMST_WORKSPACE_GIB")) s.workspaceGiB = atof(env);
		if (const char *env = std::getenv("SMST_SUB_STREAMS")) s.subStreams = atoi(env);
#ifndef SMST_NO_SWITCHES
		auto set = [](const char *name) { return std::getenv(name) != nullptr; };
		auto num = [](const char *name, int fallback) { const char *env = std::getenv(name); return env ? atoi(env) : fallback; };
		s.overlap = num("SMST_NO_OVERLAP", 0) == 0;
		s.noFuse = set("SMST_NO_FUSE");
		s.noSingleHop = set("SMST_NO_SINGLE_HOP");
		s.noAcross = set("SMST_NO_ACROSS");
		s.checkLaunches = num("SMST_CHECK_LAUNCHES", 0) != 0;
		s.noFeedFusion = num("SMST_NO_FEED_FUSION", 0);
		s.noStage = set("SMST_NO_STAGE");
		s.noAlign = set("SMST_NO_ALIGN");
		s.alignAll = set("SMST_ALIGN_ALL");
		s.continuous = set("SMST_CONTINUOUS");
		s.contWriterWave = num("SMST_CONT_WRITER_WAVE", 4) == 11 ? 11 : 4;
		s.noFastFft = set("SMST_NO_FAST_FFT");
		if (const char *env = std::getenv("SMST_FFT_TABLES")) s.fftLean = std::string(env) == "lean";
		s.feedSerial = set("SMST_FEED_SERIAL");
		s.fftTeams = num("SMST_FFT_TEAMS", 1);
		s.synthEmit = num("SMST_SYNTH_EMIT", 1);
		s.carriedEmit = num("SMST_CARRIED_EMIT", 1);
		s.vocNWide = num("SMST_VOCN_WIDE", 1);
		s.vocWide = num("SMST_VOC_WIDE", 1);
		s.vocNHalfLines = num("SMST_VOCN_HALF_LINES", 2);
#ifdef SMST_EXPERIMENTS
		s.debugMode = num("SMST_DEBUG_MODE", 0);
#endif
#endif
		return s;
	}
};

} // namespace smst
