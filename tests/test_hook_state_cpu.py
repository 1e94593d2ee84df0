"""Host logic of tools/_hook_state.py on CPU tensors (no kernel involved): the bookkeeping of the poison probe that decides
whether a convolution's output may stay un-materialised (Quantity.fuse_conv_add / skip_unread_outputs), and the statistics
gatherer's in-place check.  The GPU behaviour built on it is tests/test_gpu_conv_add_fusion.py."""
import pytest
import torch

from tools._hook_state import _AFTER_FORWARD, _DeferralProbe, _EagerStats, _HookState


class _M(object):
    """stands in for a module (the probe only uses identity)"""

    def __init__(self, name):
        self.name = name


def _learn_chain():
    conv, elt, relu, other = _M("conv3"), _M("add"), _M("relu3"), _M("conv1")
    p = _DeferralProbe()
    y = torch.arange(6.0).view(1, 2, 3)
    assert p.conv_done(conv, y) is y                       # learn mode: values untouched
    shortcut = torch.ones(1, 2, 3)
    assert p.eltwise(elt, y, shortcut) is None             # learn mode: the Eltwise runs itself ...
    assert p.pairs == {elt: conv}                          # ... and the probe knows whose output it received
    assert p.eltwise(_M("add2"), shortcut, torch.zeros(1, 2, 3)) is None and len(p.pairs) == 1   # not a convolution output
    p.candidates[conv] = (elt, relu)
    p.relu_only[other] = relu
    return p, conv, elt, relu, other, y, shortcut


def test_poison_mode_hands_out_nan_and_routes_the_real_values_privately():
    p, conv, elt, relu, other, y, shortcut = _learn_chain()
    p.mode = "poison"
    bad = p.conv_done(conv, y)
    assert bad is not y and bool(torch.isnan(bad).all()) and bad.shape == y.shape
    assert p.real(bad, elt) is y                           # the designated Eltwise gets the real tensor ...
    assert p.real(bad, relu) is None and p.real(bad, _M("x")) is None        # ... nobody else does
    assert p.real(y, elt) is None                          # (only the poisoned object is a key)
    s_bad = p.eltwise(elt, bad, shortcut)                  # the sum, poisoned again, real values kept for the ReLU
    assert bool(torch.isnan(s_bad).all())
    assert torch.equal(p.relu(relu, s_bad), torch.relu(y + shortcut))
    assert p.relu(_M("other relu"), s_bad) is None         # another reader of the sum sees the poison
    assert p.eltwise(_M("another add"), bad, shortcut) is None               # another Eltwise reads the poisoned tensor itself
    # operand order does not matter
    bad2 = p.conv_done(conv, y)
    s2 = p.eltwise(elt, shortcut, bad2)
    assert torch.equal(p.relu(relu, s2), torch.relu(shortcut + y))
    # a convolution that is not a candidate keeps its output
    plain = torch.zeros(2)
    assert p.conv_done(_M("conv2"), plain) is plain


def test_relu_only_chain_and_poisoned_keys():
    p, conv, elt, relu, other, y, shortcut = _learn_chain()
    p.mode = "poison"
    z = torch.tensor([-1.0, 2.0])
    bad = p.conv_done(other, z)
    assert bool(torch.isnan(bad).all()) and torch.equal(p.relu(relu, bad), torch.tensor([0.0, 2.0]))
    assert p.relu(_M("r"), bad) is None
    p.keys.update({conv: "Conv2d_8", elt: "Eltwise_10", other: "Conv2d_4", relu: "ReLU_11", _M("fc"): "Linear_60"})
    assert p.poisoned_keys() == {"Conv2d_8", "Eltwise_10", "Conv2d_4"}      # what the proof must NOT compare


def test_learn_mode_relu_is_a_no_op_and_ids_are_not_trusted_alone():
    p = _DeferralProbe()
    assert p.relu(_M("r"), torch.zeros(1)) is None
    conv = _M("c")
    y = torch.zeros(3)
    p.conv_done(conv, y)
    key = id(y)
    del y
    imposter = torch.ones(3)
    p.conv_out[id(imposter)] = p.conv_out.pop(key)         # a recycled id: the stored object is what counts
    assert p.eltwise(_M("e"), imposter, torch.zeros(3)) is None and not p.pairs


def test_hook_state_defaults_and_slots():
    st = _HookState()
    assert st.fuse_stat == "max" and st.keep_feats is True and st.keep_names is None
    assert st.deferred == {} and st.defer_ok == {} and st.relu_only_ok == set() and st.poison is None
    with pytest.raises(AttributeError):
        st.not_a_field = 1                                  # (a typo in a field name must not create a new one silently)


def test_eager_stats_groups_flushes_and_notices_in_place_writes():
    seen = []
    e = _EagerStats(lambda d: seen.append(dict(d)), 0)
    a, b = torch.zeros(4), torch.ones(4)
    e.add("a", a)
    e.add("b", b)
    assert [list(d) for d in seen] == [["a"], ["b"]]       # limit 0: one launch per tensor
    seen.clear()
    e = _EagerStats(lambda d: seen.append(dict(d)), _AFTER_FORWARD)
    e.add("a", a)
    e.note("c", torch.zeros(1))
    assert not seen and not e.modified()
    e.flush({"kept": b})
    assert [sorted(d) for d in seen] == [["a", "kept"]]
    e = _EagerStats(lambda d: None, _AFTER_FORWARD)
    e.add("a", a)
    a.add_(1.0)                                            # an in-place consumer after the hook
    assert e.modified()
    with pytest.raises(RuntimeError, match="modified in place"):
        e.flush()
    e = _EagerStats(lambda d: None, _AFTER_FORWARD)
    e.retain = False
    e.note("x", a)
    assert e.seen == []                                    # nothing held when no cache is wanted
