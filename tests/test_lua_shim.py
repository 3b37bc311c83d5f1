"""bindings/rau.lua cannot be executed here (no LuaJIT in the image): static checks that catch
the mistakes a run would -- unbalanced blocks, calls to C symbols the cdef does not declare,
clone methods the reference's loops use that the shim does not define."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lua_source():
    return open(os.path.join(ROOT, "bindings", "rau.lua")).read()


def strip_comments_and_strings(src):
    src = re.sub(r"--\[\[.*?\]\]", " ", src, flags=re.S)          # block comments
    src = re.sub(r"ffi\.cdef\[\[.*?\]\]", " ", src, flags=re.S)   # the C declarations
    src = re.sub(r"--[^\n]*", " ", src)
    src = re.sub(r"'(?:\\.|[^'\\])*'", "''", src)
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)
    return src


def test_blocks_balance():
    code = strip_comments_and_strings(lua_source())
    toks = re.findall(r"\b(function|if|for|while|repeat|until|do|end|then|elseif)\b", code)
    depth = 0
    pending_do = 0      # `for ... do` / `while ... do`: the `do` belongs to the loop header
    for t in toks:
        if t in ("function", "if", "repeat"):
            depth += 1
        elif t in ("for", "while"):
            depth += 1
            pending_do += 1
        elif t == "do":
            if pending_do:
                pending_do -= 1
            else:
                depth += 1
        elif t in ("end", "until"):
            depth -= 1
            assert depth >= 0, "more block closers than openers"
    assert depth == 0, f"unbalanced blocks: depth {depth} at end of file"
    for o, c in ("()", "{}", "[]"):
        assert code.count(o) == code.count(c), f"unbalanced {o}{c}"


def test_every_c_call_is_declared_in_the_cdef():
    src = lua_source()
    cdef = "\n".join(re.findall(r"ffi\.cdef\[\[(.*?)\]\]", src, flags=re.S))
    declared = set(re.findall(r"\b(rau_[a-z_0-9]+)\s*\(", cdef))
    used = set(re.findall(r"\bC\.(rau_[a-z_0-9]+)", strip_comments_and_strings(src)))
    assert used, "no C calls found"
    assert not (used - declared), f"called but not declared: {sorted(used - declared)}"


def test_surface_the_reference_loops_need():
    """Methods feval's own loops call on modules and tensors (SS:319-347, 449-461, 479-492,
    518-526, 565-593): present in the shim."""
    src = lua_source()
    for method in ("RAU:training", "RAU:evaluate", "RAU:getParameters", "RAU:cuda", "RAU:clone",
                   "RAU:updateParameters", "RAU:zeroGradParameters", "RAU:forward", "RAU:backward",
                   "Tensor:add", "Tensor:copy", "Tensor:zero", "Tensor:max", "Tensor:sum",
                   "Tensor:row", "Tensor:selectRows", "Tensor:float", "IntTensor:eqSum"):
        assert re.search(r"function\s+" + re.escape(method) + r"\b", src), method
    assert "Tensor.__newindex" in src and "Tensor.__index" in src    # t[k] and t[k] = row
    for kind in ("embed", "rnn", "multimodal", "criterion"):
        assert f"kind == '{kind}'" in src
