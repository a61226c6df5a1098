"""tools/check_async_isa.py on hand-written ISA: the build relies on it to refuse a kernel in which the compiler touches a register
whose inline-assembly load is still in flight, so the checker itself gets known-good and known-bad inputs."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_async_isa", os.path.join(ROOT, "tools", "check_async_isa.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def run(text, tmp_path):
    p = tmp_path / "k.s"
    p.write_text(text)
    out = []
    for name, body in chk.functions(str(p), ["kTest"]):
        out += chk.check_function(name, body, str(p))
    return out


HEAD = "_Z5kTestv:\n"
TAIL = "\ts_endpgm\n.Lfunc_end0:\n"
ASM_LOAD = "\t;;#ASMSTART\n\tglobal_load_dwordx2 v[10:11], v[2:3], off\n\t;;#ASMEND\n"


def test_clean_use_after_wait(tmp_path):
    assert run(HEAD + ASM_LOAD + "\ts_waitcnt vmcnt(0)\n\tv_add_f32_e32 v1, v10, v11\n" + TAIL, tmp_path) == []


def test_copy_of_a_register_in_flight_is_reported(tmp_path):
    found = run(HEAD + ASM_LOAD + "\tv_mov_b32_e32 v20, v10\n\ts_waitcnt vmcnt(0)\n" + TAIL, tmp_path)
    assert len(found) == 1 and "v_mov_b32_e32 v20, v10" in found[0]


def test_counted_wait_retires_in_order(tmp_path):
    # two loads, vmcnt(1): the older one has landed, the younger one has not
    two = ASM_LOAD + "\t;;#ASMSTART\n\tglobal_load_dwordx2 v[12:13], v[2:3], off\n\t;;#ASMEND\n\ts_waitcnt vmcnt(1)\n"
    assert run(HEAD + two + "\tv_mov_b32_e32 v20, v10\n\ts_waitcnt vmcnt(0)\n" + TAIL, tmp_path) == []
    assert len(run(HEAD + two + "\tv_mov_b32_e32 v20, v12\n\ts_waitcnt vmcnt(0)\n" + TAIL, tmp_path)) == 1


def test_compiler_loads_only_count(tmp_path):
    # a load the compiler tracks (outside ASMSTART/ASMEND) is its own business, but it takes its place in the order
    text = HEAD + ASM_LOAD + "\tglobal_load_dword v30, v[2:3], off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v20, v10\n\tv_mov_b32_e32 v31, v30\n\ts_waitcnt vmcnt(0)\n" + TAIL
    assert run(text, tmp_path) == []
    text = HEAD + ASM_LOAD + "\tglobal_store_dword v[2:3], v5, off\n\ts_waitcnt vmcnt(2)\n\tv_mov_b32_e32 v20, v10\n\ts_waitcnt vmcnt(0)\n" + TAIL
    assert len(run(text, tmp_path)) == 1


def test_in_flight_across_a_back_edge(tmp_path):
    # requested inside the loop, copied at the top of the next iteration before any wait: the hazard the first aligned producers had
    text = (HEAD + "\ts_waitcnt vmcnt(0)\n.LBB0_1:\n\tv_mov_b32_e32 v20, v10\n" + ASM_LOAD + "\ts_cbranch_scc1 .LBB0_1\n\ts_waitcnt vmcnt(0)\n" + TAIL)
    found = run(text, tmp_path)
    assert len(found) == 2 and "v_mov_b32_e32 v20, v10" in found[0] and "global_load_dwordx2 v[10:11]" in found[1]  # (the copy; the re-request into a register still in flight)


def test_functions_are_picked_by_content(tmp_path):
    """No name list: every function with an inline-assembly load is checked, whatever it is called -- and only those."""
    p = tmp_path / "k.s"
    p.write_text("_Z6kPlainv:\n\tglobal_load_dword v30, v[2:3], off\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n.Lfunc_end0:\n"
                 + "_Z9kNewAsyncILi3EEvv:\n" + ASM_LOAD + "\tv_mov_b32_e32 v20, v10\n\ts_waitcnt vmcnt(0)\n\ts_endpgm\n.Lfunc_end1:\n")
    picked = list(chk.functions(str(p)))
    assert [name for name, _ in picked] == ["_Z9kNewAsyncILi3EEvv"]
    assert len(chk.check_function(picked[0][0], picked[0][1], str(p))) == 1
