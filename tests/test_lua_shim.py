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
                   "Tensor:row", "Tensor:selectRows", "Tensor:float", "IntTensor:eqSum",
                   # the tensor statements of utils/optim_updates.lua:76-86, and owned-memory release
                   "Tensor:mul", "Tensor:addcmul", "Tensor:addcdiv", "Tensor:sqrt", "Tensor:free",
                   "IntTensor:free"):
        assert re.search(r"function\s+" + re.escape(method) + r"\b", src), method
    assert "Tensor.__newindex" in src and "Tensor.__index" in src    # t[k] and t[k] = row
    for kind in ("embed", "rnn", "multimodal", "criterion"):
        assert f"kind == '{kind}'" in src


def test_adam_has_the_reference_signature_and_owned_memory_has_finalizers():
    """SS:770-772 calls adam(x, dx, lr, alpha, beta, epsilon, state) (utils/optim_updates.lua:59):
    the shim exports that signature; device memory it allocates carries an ffi.gc finalizer that
    checks the context is still alive, and max() does not allocate per call."""
    src = lua_source()
    assert re.search(r"function\s+RAU\.adam\(x,\s*dx,\s*lr,\s*beta1,\s*beta2,\s*epsilon,\s*state\)", src)
    code = strip_comments_and_strings(src)
    assert re.search(r"ffi\.gc\(p\[0\],\s*function\(q\)\s*if\s+life\.alive", code)
    body = code[code.index("function Tensor:max"):]
    body = body[:body.index("function Tensor:selectRows")]
    assert "scratch" in body and "if not ring then" in body          # allocation only when the ring is first built
    assert body.index("if not ring then") < body.index("Tensor.new") < body.index("ring.k = ring.k % 4 + 1")
    assert "rau_dev_axpy" in code[code.index("function RAU:updateParameters"):][:400]   # plain SGD
