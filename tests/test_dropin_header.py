"""The C++ drop-in header (include/signalsmith-stretch/signalsmith-stretch.h) must accept code written against the
reference class: every public member of signalsmith-stretch.h:38-491 with the reference's argument lists, called the way
cmd/main.cpp:44-82 and the README call them (any buffer type indexable as buf[channel][index]).  Compile-only."""
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SOURCE = textwrap.dedent(r'''
    // the reference's profiling hooks (signalsmith-stretch.h:211-213, :329-331, :402-404, :420-422), defined as cmd/main-dev.cpp defines them
    static int profileCalls = 0;
    #define SIGNALSMITH_STRETCH_PROFILE_PROCESS_START(inputSamples, outputSamples) (profileCalls += (inputSamples) >= 0 && (outputSamples) >= 0)
    #define SIGNALSMITH_STRETCH_PROFILE_PROCESS_STEP(step, count) (profileCalls += (step) < (count))
    #define SIGNALSMITH_STRETCH_PROFILE_PROCESS_ENDSTEP() (++profileCalls)
    #define SIGNALSMITH_STRETCH_PROFILE_PROCESS_END() (++profileCalls)
    #include "signalsmith-stretch/signalsmith-stretch.h"
    #include <vector>
    #include <array>
    #include <functional>
    using Stretch = signalsmith::stretch::SignalsmithStretch<float>;
    int main() {
        Stretch stretch;          // :38
        Stretch seeded(12345L);   // :39
        stretch.presetDefault(2, 48000.0f);            // :63
        stretch.presetDefault(2, 48000.0f, true);
        stretch.presetCheaper(2, 48000.0f);            // :66
        stretch.presetCheaper(2, 48000.0f, false);
        stretch.configure(2, 4096, 1024);              // :71
        stretch.configure(2, 4096, 1024, true);
        int a = stretch.blockSamples() + stretch.intervalSamples() + stretch.inputLatency() + stretch.outputLatency();  // :42-47,:96-101
        bool split = stretch.splitComputation();       // :102
        stretch.reset();                               // :49
        stretch.setTransposeFactor(1.5f);              // :107
        stretch.setTransposeFactor(1.5f, 0.25f);
        stretch.setTransposeSemitones(3);              // :116
        stretch.setTransposeSemitones(3.0f, 8000.0f/48000);
        stretch.setFreqMap([](float f) { return f*1.2f; });   // :120
        stretch.setFreqMap(nullptr);
        stretch.setFormantFactor(1.1f);                // :124
        stretch.setFormantFactor(1.1f, true);
        stretch.setFormantSemitones(2);                // :129
        stretch.setFormantSemitones(2.0f, true);
        stretch.setFormantBase();                      // :133
        stretch.setFormantBase(200.0f/48000);
        std::vector<std::vector<float>> in(2, std::vector<float>(8192)), out(2, std::vector<float>(8192));
        float *inPtr[2] = {in[0].data(), in[1].data()}, *outPtr[2] = {out[0].data(), out[1].data()};
        std::array<std::vector<float>, 2> inArr{{in[0], in[1]}};
        stretch.seek(in, 1000, 1.0);                   // :140 (any indexable buffers)
        stretch.seek(inPtr, stretch.seekLength(), 0.5);   // :166
        int osl = stretch.outputSeekLength(1.0f);      // :205
        stretch.outputSeek(inArr, osl);                // :173
        stretch.process(in, 4096, out, 4096);          // :210
        stretch.process(inPtr, 1000, outPtr, 1500);
        stretch.flush(out, 512);                       // :427
        stretch.flush(outPtr, 512, 1.0f);
        bool ok = stretch.exact(in, 8192, out, 8192);  // :468
        static_assert(Stretch::version[0] == 1 && Stretch::version[1] == 3, "reference API version");   // :36
        // the reference class is a plain struct: value semantics (:34-35)
        Stretch copy(stretch);                         // copy constructor: carries the processing state
        copy = seeded;                                 // copy assignment
        Stretch moved(std::move(copy));                // move
        moved = Stretch(7L);
        std::vector<Stretch> pool(3);                  // containers of them
        pool.push_back(stretch);
        pool.emplace_back(99L);
        static_assert(std::is_copy_constructible<Stretch>::value && std::is_nothrow_move_constructible<Stretch>::value, "value semantics");
        Stretch::setDefaultDevice(0);
        // Sample = double (:34): accepted at the interface -- buffers and parameters are converted, the device computes in fp32
        signalsmith::stretch::SignalsmithStretch<double> wide(5L);
        wide.presetDefault(2, 48000.0);
        wide.setTransposeSemitones(3.0, 0.2);
        wide.setFreqMap([](double f) { return f*1.1; });
        std::vector<std::vector<double>> inD(2, std::vector<double>(4096)), outD(2, std::vector<double>(6144));
        wide.process(inD, 4096, outD, 6144);
        wide.flush(outD, 100, 1.0);
        return (a > 0 && ok && split) ? 0 : 1;
    }
''')


def test_reference_style_code_compiles(tmp_path):
    src = tmp_path / "dropin.cpp"
    src.write_text(SOURCE)
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


MOVED_FROM = textwrap.dedent(r'''
    #include "signalsmith-stretch/signalsmith-stretch.h"
    #include <cstdio>
    #include <vector>
    using Stretch = signalsmith::stretch::SignalsmithStretch<float>;
    int main() {
        Stretch a(3L);
        a.presetCheaper(1, 48000.0f);
        Stretch b(std::move(a));                 // a is moved-from: still a valid object, as the reference's struct is
        if (a.blockSamples() >= 0 && b.blockSamples() != 4800) return 1;
        Stretch c(a);                            // copying a moved-from object copies an unconfigured one
        a.configure(1, 512, 128);                // ... and it can be configured and used again
        std::vector<std::vector<float>> in(1, std::vector<float>(2048, 0.25f)), out(1, std::vector<float>(2048));
        a.process(in, 2048, out, 2048);
        c = a;                                   // copy assignment from the re-used object
        if (c.blockSamples() != 512 || a.intervalSamples() != 128) return 2;
        b = std::move(c);
        c.presetDefault(2, 44100.0f);            // moved-from by assignment, used again
        std::printf("ok %d %d\n", b.blockSamples(), c.blockSamples());
        return (b.blockSamples() == 512 && c.blockSamples() == 5292) ? 0 : 3;
    }
''')


def test_moved_from_objects_stay_usable(emu, tmp_path):
    """a moved-from drop-in object used to pass a null handle to every call.  Runs against the CPU stand-in."""
    src, exe = tmp_path / "moved.cpp", tmp_path / "moved"
    src.write_text(MOVED_FROM)
    emu_dir = os.path.join(ROOT, "tests", "emu")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L" + emu_dir, "-l:libsmst_emu.so",
                        "-Wl,-rpath," + emu_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])


PROFILE_HOOKS = textwrap.dedent(r'''
    #include <cstdio>
    #include <cstddef>
    #include <vector>
    // what a harness like cmd/main-dev.cpp:12-58 sees: per call, the (step, steps) pairs and whether every STEP was closed
    static std::vector<std::pair<size_t, size_t>> seen;
    static int opened = 0, closed = 0, started = 0, ended = 0;
    #define SIGNALSMITH_STRETCH_PROFILE_PROCESS_START(inputSamples, outputSamples) (++started, seen.clear())
    #define SIGNALSMITH_STRETCH_PROFILE_PROCESS_STEP(step, count) (++opened, seen.emplace_back(size_t(step), size_t(count)))
    #define SIGNALSMITH_STRETCH_PROFILE_PROCESS_ENDSTEP() (++closed)
    #define SIGNALSMITH_STRETCH_PROFILE_PROCESS_END() (++ended)
    #include "signalsmith-stretch/signalsmith-stretch.h"
    using Stretch = signalsmith::stretch::SignalsmithStretch<float>;
    static bool consistent() { // ONE count per call, steps 0 .. count-1 in order
        for (size_t i = 0; i < seen.size(); ++i) if (seen[i].first != i || seen[i].second != seen.size()) return false;
        return opened == closed;
    }
    int main() {
        Stretch s;
        s.configure(2, 512, 128);
        std::vector<std::vector<float>> in(2, std::vector<float>(4096)), out(2, std::vector<float>(4096));
        for (int c = 0; c < 2; ++c) for (int i = 0; i < 4096; ++i) in[c][i] = 0.3f*float((i*7 + c*13)%97)/97 - 0.15f;
        s.process(in, 100, out, 100);                          // the first block begins: plain, stereo
        if (seen.empty() || !consistent()) { std::printf("first call: %zu steps\n", seen.size()); return 1; }
        const size_t plainSteps = seen.size();
        s.process(in, 20, out, 20);                            // inside the interval: no block begins, nothing is announced
        if (!seen.empty()) { std::printf("a call without a block announced %zu steps\n", seen.size()); return 2; }
        s.setTransposeSemitones(4, 0);
        s.process(in, 300, out, 300);                          // a mapped block: more steps, and again ONE count
        if (seen.size() <= plainSteps || !consistent()) { std::printf("mapped call: %zu steps (plain %zu)\n", seen.size(), plainSteps); return 3; }
        std::printf("ok plain %zu mapped %zu calls %d/%d\n", plainSteps, seen.size(), started, ended);
        return (started == 3 && ended == 3) ? 0 : 4;
    }
''')


def test_profiling_hooks_report_one_count_per_call(emu, tmp_path):
    """STEP(step, steps) carried two different `steps` in one call, and a call without a block re-announced the last one's."""
    src, exe = tmp_path / "hooks.cpp", tmp_path / "hooks"
    src.write_text(PROFILE_HOOKS)
    emu_dir = os.path.join(ROOT, "tests", "emu")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L" + emu_dir, "-l:libsmst_emu.so",
                        "-Wl,-rpath," + emu_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
