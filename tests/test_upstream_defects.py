"""Evidence for the flag combinations sthip_render rejects with STHIP_ERR_UNSUPPORTED because the REFERENCE reads state it
never writes (api.hip: the `fail(ctx, STHIP_ERR_UNSUPPORTED, ...)` sites; DESIGN.md "Out of scope"). A rejection of that kind
is only as good as the claim about upstream it rests on, so each claim is pinned here twice:

  * against the reference's own text, where this container has it (/root/reference; skipped elsewhere — the GPU box has no
    reference): the lines the claim names must still say what the claim says. Reading source as text is study, nothing
    of it is stored here: the assertions are about which names are (not) assigned in which line ranges;
  * as arithmetic, where the defect is a formula (thread padding).

The combinations rejected for another reason — media together with eCoherentSampling (walks through volumes break the lockstep of a group), reservoir reuse on a tile shard (a whole-frame structure) — are design limits of this
library, not upstream defects, and are listed as such in DESIGN.md; nothing here covers them."""
import os
import re

import numpy as np
import pytest

REF = "/root/reference/src/Shaders"
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's sources are only present in the build container")


def _lines(rel, first, last):
    with open(os.path.join(REF, rel)) as f:
        text = f.read().split("\n")
    return text[first - 1 : last]


def _assigned(lines, name):
    """True if `name` is the target of an assignment (name = ..., name.x = ..., BF_SET(name, ...)) in these lines."""
    pat = re.compile(r"(^|[^\w.])%s(\.\w+)?\s*(=[^=]|\*=|\+=)" % re.escape(name))
    bf = re.compile(r"BF_SET\(\s*%s\b" % re.escape(name))
    return any(pat.search(ln) or bf.search(ln) for ln in lines)


@needs_reference
def test_an_environment_sample_has_no_position_and_no_normal():
    """`sample_point_on_light` (common/light.hlsli:37-152) fills `out LightSampleRecord ls`. Its environment branch (:38-48)
    writes radiance, to_light, pdf, the instance bits, dist, pdf_area_measure and material_address — and neither
    ls.position nor ls.normal. Two callers read exactly those for every sample, environment or not:
      presample_lights (bdpt.hlsl:84-99) stores ls.position / ls.normal as the presampled point  -> ePresampleLights with an
        environment stores an undefined position (api.hip: "upstream leaves the presampled environment direction unset");
      sample_photons (bdpt.hlsl:101-147) starts the light path at ray_offset(ls.position, ls.normal) in a direction built
        around ls.normal                                                  -> eConnectToViews / eConnectToLightPaths with
        an environment trace light paths from an undefined origin (api.hip: "light subpaths with an environment").
    The same record is what the NEE reservoir stores (sample_Le, path.hlsli:141-164: r.position = ls.position), which is why
    reservoir reuse with an environment reads an undefined point back."""
    env_branch = _lines("common/light.hlsli", 38, 48)
    assert any("sample environment" in ln for ln in env_branch), "the environment branch moved: re-anchor this test"
    for written in ("ls.radiance", "ls.dist", "ls.pdf_area_measure", "ls.material_address", "ls.instance_primitive_index"):
        assert _assigned(env_branch, written), written
    assert any("ls.to_light" in ln for ln in env_branch)  # (an out argument of env.sample)
    assert not _assigned(env_branch, "ls.position")
    assert not _assigned(env_branch, "ls.normal")
    # the emitter branch does write them (so the names are right)
    emitter_branch = _lines("common/light.hlsli", 49, 152)
    assert _assigned(emitter_branch, "ls.position") and _assigned(emitter_branch, "ls.normal")
    # ... and the callers read them unconditionally
    presample = "\n".join(_lines("kernels/renderers/bdpt.hlsl", 84, 99))
    assert "l.position = ls.position" in presample and "pack_normal_octahedron(ls.normal)" in presample and "is_environment" in presample
    photons = "\n".join(_lines("kernels/renderers/bdpt.hlsl", 101, 147))
    assert "path._isect.sd.position = ls.position" in photons and "ray_offset(ls.position, ls.normal)" in photons
    assert "is_environment" not in photons  # no special case for an environment sample
    sample_le = "\n".join(_lines("common/path.hlsli", 141, 164))
    assert "r.position = ls.position" in sample_le


@needs_reference
def test_the_light_power_distribution_is_built_from_an_unwritten_table():
    """eSampleLightPower: `build_distribution(dist, pmf, cdf)` (dist2.h:63-78) accumulates `pmf[i]` into the cdf before
    anything has written pmf — its input `dist` is never read — so the light-power tables the flag selects (light.hlsli:26-28)
    hold whatever the allocation held. There is nothing to be identical to: the flag is rejected."""
    body = _lines("dist2.h", 63, 78)
    text = "\n".join(body)
    assert "build_distribution(const vector<float>& dist" in text
    first_loop = "\n".join(body[1:5])
    assert "cdf[i + 1] = cdf[i] + pmf[i]" in first_loop
    # between the signature and that loop nothing assigns pmf, and `dist` is only used for its size
    assert not _assigned(body[:4], "pmf[i]") and not _assigned(body[:4], "pmf")
    uses_of_dist = [ln for ln in body if re.search(r"\bdist\b", ln) and "build_distribution" not in ln]
    assert all("dist.size()" in ln for ln in uses_of_dist), uses_of_dist


@needs_reference
def test_the_padding_threads_of_sample_photons_are_not_masked():
    """sample_photons returns only for `path_index >= gLightPathCount` (bdpt.hlsl:104-105); without eRemapThreads the path
    index is pixel_coord.y * gOutputExtent.x + pixel_coord.x (bdpt_util.hlsli:76-83), and nothing compares pixel_coord.x
    with the width. The dispatch covers whole 8 x 4 groups, so when the width is not a multiple of 8 the threads of the
    padding columns carry the indices of the NEXT row's first pixels: two threads write the same light-vertex slots."""
    photons = "\n".join(_lines("kernels/renderers/bdpt.hlsl", 101, 107))
    assert "if (path_index >= gLightPathCount) return;" in photons
    assert "gOutputExtent" not in photons  # no test of index.x against the width
    mapping = "\n".join(_lines("common/bdpt_util.hlsli", 76, 83))
    assert "return pixel_coord.y*gOutputExtent.x + pixel_coord.x;" in mapping


@pytest.mark.parametrize("width,races", [(96, False), (100, True), (161, True), (1920, False)])
def test_the_padding_columns_collide_with_the_next_row(width, races):
    """The arithmetic of the claim above: over the padded dispatch (ceil(W / 8) * 8 columns), the map (x, y) -> y * W + x is
    injective exactly when W is a multiple of 8. This is the condition api.hip rejects eConnectToLightPaths without
    eRemapThreads on ("upstream's padding threads race on the vertex slots of the next row")."""
    height = 12
    padded = (width + 7) // 8 * 8
    x, y = np.meshgrid(np.arange(padded), np.arange(height))
    index = (y * width + x).ravel()
    live = index < width * height  # gLightPathCount = W * H without the vertex cache (BDPT.cpp:469-470)
    collisions = np.unique(index[live]).size != int(live.sum())
    assert collisions == races


@needs_reference
def test_every_image_read_of_the_path_names_its_level():
    """The reference's sampler asks for 8x anisotropic filtering (BDPT.cpp:130-133), the library models trilinear filtering over a
    mip chain. Not a gap: anisotropic filtering works on the derivatives of an implicit-level read, and every image read of the
    path tracer's shaders is `SampleLevel` with the level worked out from the ray cone (image_value.h:81-97; environment.h:53,74;
    the alpha test in intersection.hlsli) — no derivative ever reaches the sampler, so its maxAnisotropy has nothing to act on.
    The one implicit-level `.Sample(` of the tree is the GUI's rasteriser (kernels/renderers/raster.hlsl), out of scope."""
    explicit, implicit = [], []
    for root, _, files in os.walk(REF):
        for f in files:
            if not f.endswith((".h", ".hlsli", ".hlsl", ".slang")):
                continue
            rel = os.path.relpath(os.path.join(root, f), REF)
            text = open(os.path.join(root, f), errors="replace").read()
            if re.search(r"\.SampleLevel\(", text):
                explicit.append(rel)
            if re.search(r"\.(Sample|SampleGrad|SampleBias|SampleCmp)\(", text):
                implicit.append(rel)
    assert "image_value.h" in explicit and "environment.h" in explicit and os.path.join("common", "intersection.hlsli") in explicit
    assert implicit == [os.path.join("kernels", "renderers", "raster.hlsl")], implicit
