"""The generated robot tables (steppingstone_amd/csrc/ss_model_tables.hpp for the kernels, oracle/ss_model_tables.h for the oracle) must be
what tools/gen_model_tables.py produces from steppingstone_amd/model.py + identified_*.json RIGHT NOW: since round 5 the specification's
numbers live in data files, and a stale table would silently put the kernels, the oracle and the numpy restatements on different robots."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_committed_tables_equal_the_generator_output():
    import gen_model_tables as gen
    from steppingstone_amd import model
    for kind, _, _ in gen.KINDS:
        gen.assert_mirror_symmetric(model.build(kind))
    assert open(os.path.join(ROOT, "steppingstone_amd", "csrc", "ss_model_tables.hpp")).read() == gen.gen_hpp()
    assert open(os.path.join(ROOT, "oracle", "ss_model_tables.h")).read() == gen.gen_h()


def test_identified_numbers_are_what_the_specification_evaluates():
    from steppingstone_amd import model
    for kind in ("walker3d", "mike"):
        ident = model.identified(kind)
        assert ident, "steppingstone_amd/identified_%s.json is missing" % kind
        m, prior = model.build(kind), model.build(kind, use_identified=False)
        assert abs(m["friction"] - ident["friction"]) < 1e-12 and m["mass"].sum() != prior["mass"].sum()
    ec = model.env_constants()
    assert ec["stone_plank_half_length"] == 0.30 and 0.30 <= ec["stone_plank_half_width"] <= 0.60      # a plank that cannot overlap its neighbour


def test_the_identified_robots_pass_the_plausibility_assertions():
    """the compiled numbers describe a robot -- nominal pose strictly inside the joint ranges (the reset clip never pins
    a joint), sole = the foot box's bottom face 4-10 cm below the ankle, friction and mass multipliers inside their stated bounds
    (tools/gen_model_tables.py: assert_plausible)."""
    import numpy as np
    import gen_model_tables as gen
    from steppingstone_amd import model
    for kind in ("walker3d", "mike"):
        assert gen.assert_plausible(kind) is True, "%s: no round-6 identified file" % kind
        m = model.build(kind)
        lo, hi = m["range"][:, 0] + 0.02, m["range"][:, 1] - 0.02                 # PHYSICS.md 7: q = clip(q0 + 0.05 (2u - 1), lo + 0.02, hi - 0.02)
        assert (m["q0"] - 0.05 >= lo - 1e-9).all() and (m["q0"] + 0.05 <= hi + 1e-9).all()
        c = m["corners"]
        assert np.ptp(c[:, 2]) == 0 and -0.10 <= c[0, 2] <= -0.04
