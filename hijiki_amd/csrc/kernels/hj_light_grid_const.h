// The numbers the light-shaft grid's builder (api/light_grid.cpp, host) and its reader (hj_shade.h, device) have to agree on.
#pragma once

namespace hj {

// A proof of a cell on a mesh or in a corner (api/light_grid.cpp, "bundle proofs") starts from "the hit point lies on one of the
// cell's shapes".  The reference's hit point is p = o + t d with t from a float shape test whose error grows with 1 / |cos(d, n)|: a
// ray that met its shape T at a grazing angle leaves p anywhere ALONG the ray, off T sideways, and the float (u, v) test accepts
// rays that pass T's edge on the outside by a similar amount.  So such a proof is used only after the shade stage has CHECKED where
// this hit point is (hj_shade.h hit_point_on_its_shape; per shape: unit normal n, vertex a, margin delta from the builder):
//   * |d.n| >= kLightGridSinIn |d|                       - not grazing (what delta's derivation assumes);
//   * (|n.(p - a)| + e) |d| <= kLightGridSlide |d.n|     - p = X + s d with X on T's plane and |s d| <= kLightGridSlide: how far p
//                                                          slid along the ray, measured on the hit point itself (e: what the float
//                                                          evaluation of n.(p - a) can lose, kLightGridUlp5 x the 1-norm of p - a);
//   * min(u, v, 1 - u - v) >= delta (a quad: u, v, 1 - u, 1 - v) - X inside T for certain, whatever rounding did to (u, v).
// Together: p within kLightGridSlide of T (the builder's sigma is 5e-6).
// Planar cells need none of this.
constexpr float kLightGridSinIn = 0.1f;
constexpr float kLightGridSlide = 4.9e-6f;
constexpr float kLightGridUlp5 = 3e-7f;                  // 5 x 2^-24, rounded up

// DeviceScene::lg_res: cells per axis (<= 256) and whether the records for that check lie behind the cells
constexpr unsigned kLightGridResMask = 0xFFFFu, kLightGridHasRecords = 0x80000000u;

}  // namespace hj
