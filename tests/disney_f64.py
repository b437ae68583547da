"""A THIRD statement of the reference's material arithmetic, in float64 numpy, written from the published formulas the
HLSL implements — not from the oracle's or the kernel's lines (which share 28 % of their text, so a mis-transcribed
term would pass every GPU-vs-oracle test). Vectorised over N direction pairs; names follow the papers, not the code.

  * microfacet normal distribution: anisotropic GGX / GTR2, D = 1 / (pi ax ay (hx^2/ax^2 + hy^2/ay^2 + hz^2)^2)
    (Burley 2012 eq. 13; materials/disney_material.hlsli:4-10)
  * Smith masking for it: G1(w) = 2 / (1 + sqrt(1 + (ax^2 wx^2 + ay^2 wy^2) / wz^2))          (Heitz 2014; :11-17)
  * exact unpolarised dielectric Fresnel from the cosine of incidence and eta                     (microfacet.h:33-53)
  * Schlick's approximation                                                                        (microfacet.h:22-27)
  * the clearcoat lobe: GTR1 with the fixed 0.25 roughness masking and eta = 1.5                  (Burley 2012 eq. 4; disney_clearcoat.hlsli)
  * Burley's diffuse with the Hanrahan-Krueger subsurface blend                                    (Burley 2015; disney_diffuse.hlsli:1-17)
  * the lobe mix of DisneyMaterial::eval                                                            (disney_material.hlsli:141-200)
and, from path.hlsli: the power-heuristic weight (:8-15), the dVC recurrence (:31-38), the shading-normal correction
(:67-98). Test infrastructure: used by tests/test_disney_f64.py only."""
import numpy as np

PI = np.pi


def _dot(a, b):
    return (a * b).sum(-1)


def _normalize(v):
    return v / np.sqrt(_dot(v, v))[..., None]


def ggx_d(h, ax, ay):
    t = (h[..., 0] / ax) ** 2 + (h[..., 1] / ay) ** 2 + h[..., 2] ** 2
    return 1.0 / (PI * ax * ay * t * t)


def smith_g1(w, ax, ay):
    tan2 = ((w[..., 0] * ax) ** 2 + (w[..., 1] * ay) ** 2) / w[..., 2] ** 2
    return 2.0 / (1.0 + np.sqrt(1.0 + tan2))  # = 1 / (1 + Lambda), Lambda = (sqrt(1 + tan2) - 1) / 2


def fresnel_dielectric(cos_i, eta):
    """eta = n_transmitted / n_incident; cos_i may be negative; total internal reflection -> 1."""
    cos_i = np.asarray(cos_i, np.float64)
    eta = np.broadcast_to(np.asarray(eta, np.float64), cos_i.shape)
    sin2_t = (1.0 - cos_i * cos_i) / (eta * eta)
    tir = sin2_t > 1.0
    cos_t = np.sqrt(np.where(tir, 0.0, 1.0 - sin2_t))
    ci = np.abs(cos_i)
    rs = (ci - eta * cos_t) / (ci + eta * cos_t)
    rp = (eta * ci - cos_t) / (eta * ci + cos_t)
    return np.where(tir, 1.0, 0.5 * (rs * rs + rp * rp))


def schlick(f0, cos_theta):
    return f0 + (1.0 - f0) * np.maximum(1.0 - cos_theta, 0.0) ** 5


def gtr1(alpha, hz):
    a2 = alpha * alpha
    return (a2 - 1.0) / (PI * np.log(a2) * (1.0 + (a2 - 1.0) * hz * hz))


def clearcoat_g1(w):
    return smith_g1(w, 0.25, 0.25)


def burley_diffuse(base_color, roughness, subsurface, wi, wo):
    """f * |cos theta_o| of the Disney diffuse lobe with its subsurface approximation."""
    h = _normalize(wi + wo)
    hdotwo = np.abs(_dot(h, wo))
    ci, co = np.abs(wi[..., 2]), np.abs(wo[..., 2])
    fss90 = roughness * hdotwo * hdotwo
    fd90 = 0.5 + 2.0 * fss90
    a, b = (1.0 - ci) ** 5, (1.0 - co) ** 5
    base = (1.0 + (fd90 - 1.0) * a) * (1.0 + (fd90 - 1.0) * b) / PI
    ss = 1.25 / PI * ((1.0 + (fss90 - 1.0) * a) * (1.0 + (fss90 - 1.0) * b) * (1.0 / (ci + co) - 0.5) + 0.5)
    return base_color[None, :] * (((1.0 - subsurface) * base + subsurface * ss) * co)[:, None]


def disney_eval(params, wi, wo, adjoint=False):
    """params: dict(base_color(3), emission, metallic, roughness, anisotropic, subsurface, clearcoat, clearcoat_gloss,
    transmission, eta). wi, wo: (N, 3) unit vectors in the shading frame (z = normal). Returns f (N, 3) INCLUDING
    |cos theta_o|, pdf_fwd, pdf_rev (solid angle)."""
    wi, wo = np.asarray(wi, np.float64), np.asarray(wo, np.float64)
    n = wi.shape[0]
    f, pf, pr = np.zeros((n, 3)), np.zeros(n), np.zeros(n)
    if params["emission"] > 0:
        return f, pf, pr
    base = np.asarray(params["base_color"], np.float64)
    metallic, trans = params["metallic"], params["transmission"]
    w_diffuse, w_metal, w_glass, w_coat = (1 - trans) * (1 - metallic), metallic, trans * (1 - metallic), 0.25 * params["clearcoat"]
    aspect = np.sqrt(1.0 - 0.9 * params["anisotropic"])
    alpha = params["roughness"] ** 2
    ax, ay = max(1e-4, alpha / aspect), max(1e-4, alpha * aspect)
    eta = np.where(wi[:, 2] < 0, 1.0 / params["eta"], params["eta"])
    transmit = wi[:, 2] * wo[:, 2] < 0
    h = _normalize(np.where(transmit[:, None], wi + wo * eta[:, None], wi + wo))
    h = np.where((h[:, 2] * wi[:, 2] < 0)[:, None], -h, h)
    hi, ho = _dot(h, wi), _dot(h, wo)
    D, Gi, Go = ggx_d(h, ax, ay), smith_g1(wi, ax, ay), smith_g1(wo, ax, ay)
    F = fresnel_dielectric(hi, eta)
    ci, co = wi[:, 2], wo[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        if w_glass > 0:
            # refraction: Walter et al. 2007 eq. 21 with the adjoint's missing 1 / eta^2, times |cos_o|
            denom = (hi + eta * ho) ** 2
            ft = np.sqrt(base)[None, :] * (((1.0 / eta**2 if adjoint else 1.0) * (1 - F) * D * Gi * Go * np.abs(ho * hi)) / (np.abs(ci) * denom))[:, None]
            pt_f = (1 - F) * D * Gi * np.abs((eta * eta * ho / denom) * hi / ci)
            Fr = fresnel_dielectric(ho, 1.0 / eta)
            denom_r = (ho + hi / eta) ** 2
            pt_r = (1 - Fr) * D * Go * np.abs(((1.0 / eta**2) * hi / denom_r) * ho / co)
            fr_ = base[None, :] * ((F * D * Gi * Go) / (4 * np.abs(ci)))[:, None]
            pr_f = F * D * Gi / (4 * np.abs(ci))
            pr_r = fresnel_dielectric(ho, eta) * D * Go / (4 * np.abs(co))
            f += w_glass * np.where(transmit[:, None], ft, fr_)
            pf += w_glass * np.where(transmit, pt_f, pr_f)
            pr += w_glass * np.where(transmit, pt_r, pr_r)
        refl = ~transmit
        if w_metal > 0:
            fm = base[None, :] * schlick(base[None, :], np.abs(ho)[:, None]) * (D * Gi * Go / (4 * np.abs(ci)))[:, None]
            f += w_metal * np.where(refl[:, None], fm, 0.0)
            pf += w_metal * np.where(refl, D * Gi / (4 * np.abs(ci)), 0.0)
            pr += w_metal * np.where(refl, D * Go / (4 * np.abs(co)), 0.0)
        if w_coat > 0:
            ag = (1 - params["clearcoat_gloss"]) * 0.1 + params["clearcoat_gloss"] * 0.001
            Dc = gtr1(ag, h[:, 2])
            Fc = schlick(((1.5 - 1) / (1.5 + 1)) ** 2, ho)
            fc = Fc * Dc * clearcoat_g1(wi) * clearcoat_g1(wo) / (4 * np.abs(ci))
            f += w_coat * np.where(refl, fc, 0.0)[:, None]
            pf += w_coat * np.where(refl, Dc * np.abs(h[:, 2]) / (4 * np.abs(ho)), 0.0)
            pr += w_coat * np.where(refl, Dc * np.abs(h[:, 2]) / (4 * np.abs(hi)), 0.0)
        if w_diffuse > 0:
            fd = burley_diffuse(base, params["roughness"], params["subsurface"], wi, wo)
            f += w_diffuse * np.where(refl[:, None], fd, 0.0)
            pf += w_diffuse * np.where(refl, np.abs(co) / PI, 0.0)
            pr += w_diffuse * np.where(refl, np.abs(ci) / PI, 0.0)
    return f, pf, pr


# ---- path.hlsli helpers ----
def power_heuristic(a, b):
    return a * a / (a * a + b * b)


def connection_dvc(dvc, pdfa_rev, prev_pdfa_fwd, specular):
    return ((0.0 if specular else 1.0) + dvc * pdfa_rev**2) / prev_pdfa_fwd**2


def shading_normal_correction(ndotin, ndotout, ngdotin, ngdotout, ngdotns, shadow_fix=False, adjoint=False):
    ndotin, ndotout, ngdotin, ngdotout, ngdotns = (np.asarray(v, np.float64) for v in (ndotin, ndotout, ngdotin, ngdotout, ngdotns))
    leak = np.sign(ngdotout * ngdotin) != np.sign(ndotin * ndotout)
    g = np.ones_like(ndotin)
    with np.errstate(divide="ignore", invalid="ignore"):
        if shadow_fix:  # Chiang et al. 2019, "Taming the shadow terminator"
            x = np.minimum(1.0, np.abs(ngdotin / (ndotin * ngdotns) if adjoint else ngdotout / (ndotout * ngdotns)))
            g = -(x**3) + x**2 + x
        if adjoint:  # Veach 5.3.2: |w_o . n_g| |w_i . n_s| / (|w_o . n_s| |w_i . n_g|)
            num, den = ngdotout * ndotin, ndotout * ngdotin
            g = np.where(np.abs(den) > 1e-5, g * np.abs(num / den), g)
    return np.where(leak, 0.0, g)
