// ab_physics.hpp — per-cell device physics of the bulk air-sea flux engine (HIP, gfx950).
//
// One wavefront lane owns one grid cell; everything below is scalar per lane and lives in
// VGPRs.  Written from the physics (Fairall 1996/2003, Edson 2013, Large & Yeager 2004/2008,
// IFS Cy40r1 ch.3, Zeng & Beljaars 2005, Andreas 2015, Grachev 2007) with the observable
// behaviour of AeroBulk as the contract: each function cites the reference file:line
// (brodeau/aerobulk `src/`) whose results it must reproduce to <=1e-10 relative.
//
// Differences w.r.t. the reference that are deliberate (not observable at 1e-10):
//   * the reference evaluates BOTH the stable and unstable psi branches and blends them with
//     a 0/1 weight; here the lane picks its branch (exact: the weight is exactly 0 or 1);
//   * loop invariants (alpha_sw(SST), warm-layer constants, logs of heights) are hoisted;
//   * `x**y` is strength-reduced (ab_math.hpp); a/exp(b) is a*exp(-b); sums of logs are fused;
//   * divisions go through Mth<R>::div (rcp + Newton, ab_fastmath.hpp), divisions by literals become
//     multiplications by the literal's reciprocal; where only LOG(z0t) is needed it is formed in the
//     log domain (no exp/log round trip).
#pragma once
#include "ab_math.hpp"

// code-region marks for the host cost model (tools/costmodel.cpp); nothing in the kernels
#ifndef AB_REGION
#define AB_REGION(name)
#endif
#ifndef AB_COUNT            // tools/costmodel.cpp: issue slots of a piece of plain double arithmetic its instrumented type cannot see
#define AB_COUNT(name, slots)
#endif

namespace ab {

template <class R> struct K {  // constants, mod_const.f90:38-114
    static constexpr R grav = R(9.8);
    static constexpr R rpi = R(3.141592653589793);
    static constexpr R roce_alb0 = R(0.066);
    static constexpr R emiss_w = R(0.98);
    static constexpr R stefan = R(5.67E-8);
    static constexpr R rt0 = R(273.15);
    static constexpr R rCp0_w = R(4190.);
    static constexpr R rho0_w = R(1025.);
    static constexpr R rnu0_w = R(1.e-6);
    static constexpr R rk0_w = R(0.6);
    static constexpr R rCp_dry = R(1005.0);
    static constexpr R rCp_vap = R(1860.0);
    static constexpr R R_dry = R(287.05);
    static constexpr R R_gas = R(8.314510);
    static constexpr R rmm_dryair = R(28.9647e-3);
    static constexpr R rmm_water = R(18.0153e-3);
    static constexpr R rLevap = R(2.46e+6);
    static constexpr R vkarmn = R(0.4);
    static constexpr R vkarmn2 = R(0.4 * 0.4);
    static constexpr R rdct_qsat_salt = R(0.98);
    static constexpr R z0_sea_max = R(0.0025);
    static constexpr R Cx_min = R(0.1E-3);
    static constexpr R rdt = R(3600.);   // mod_const.f90:32
    static constexpr R gdept1 = R(1.);   // mod_const.f90:31
    static constexpr R rpoiss_dry = R(287.05 / 1005.0);
    static constexpr R rgamma_dry = R(9.8 / 1005.0);
    static constexpr R reps0 = R(287.05 / 461.495);
    static constexpr R one_m_reps0 = R(1. - 287.05 / 461.495);
    static constexpr R rctv0 = R(461.495 / 287.05 - 1.);
    // -16*9.80665*rho0_w*rCp0_w*rnu0_w^3/rk0_w^2, mod_const.f90:109
    static constexpr R rcst_cs = R(-16. * 9.80665 * 1025. * 4190. * 1.e-6 * 1.e-6 * 1.e-6 / (0.6 * 0.6));
    static constexpr R sq_radrw = R(0.034215956910732065);  // sqrt(1.2/1025.), mod_const.f90:112
    static constexpr R inv_vk = R(2.5);                    // 1/0.4 (exact in binary up to 1 ulp of 0.4)
};

// ---------------------------------------------------------------- thermodynamics (mod_phymbl.f90)
// e_sat_sclr :777-800 — Goff (1957), T floored at 180 K:
//   e_s = 100 * 10^A(T),  A = 10.79574(1-T0/T) - 5.028 log10(T/T0) + 1.50475e-4 (1 - 10^(-8.2969(T/T0-1)))
//                             + 0.42873e-3 (10^(4.76955(1-T0/T)) - 1) + 0.78614
// The formula is evaluated 13x per cell in the skin configuration (3 exp10 + log10 + a division each time), always at
// sea-surface-like temperatures.  On 265 K <= T <= 312 K the exponent A(T) is therefore taken from its degree-14
// near-minimax polynomial in x = (T-288.5)/23.5 (60-digit Chebyshev fit of the formula above WITH ITS LITERALS AS THE DOUBLES THEY
// ARE AT RUN TIME — rt0 = 273.15 is 2.3e-14 below 273.15, seven ulp of e_sat: the first fit used the decimal values and sat
// 1.3e-15 below the reference — tools/gen_poly.py; |dA| <= 9.2e-17, i.e. below the rounding of the direct evaluation); outside
// that range (polar air, masked cells) the formula itself is evaluated.
AB_TAB double kGoffA[fm::ab_pad4(15)] = {
    1.2415921763001392, 0.6554537644583072, -0.060421505145516974, 0.005206186792148093,
    -0.00043899022431827153, 4.018144021450923e-05, -4.2993037656581925e-06, 5.668664093041562e-07,
    -8.717077199072552e-08, 1.3977187155429884e-08, -2.167668513463531e-09, 3.1479645979515667e-10,
    -4.254827005592515e-11, 5.607982060924723e-12, -6.519141017660857e-13};
__device__ __forceinline__ double goff_poly(double x)
{
    return fm::horner_coefs<15, fm::kC_Goff13>(kGoffA, x);
}
__device__ __forceinline__ float goff_poly(float x)
{
    // fp32: the first 9 terms leave < 2e-8 in A; the coefficients as float LITERALS (read from the fp64 table each of them cost a
    // v_cvt_f32_f64, a full issue slot, i.e. more than the polynomial's own FMAs)
    constexpr float c[9] = {1.2415921763001392f, 0.6554537644583072f, -0.060421505145516974f, 0.005206186792148093f, -0.00043899022431827153f, 4.018144021450923e-05f, -4.2993037656581925e-06f, 5.668664093041562e-07f, -8.717077199072552e-08f};
    float p = c[8];
#pragma unroll
    for (int i = 7; i >= 0; --i) p = __builtin_fmaf(p, x, c[i]);
    return p;
}
// Horner evaluation of a constant-memory coefficient table (coefficients fetched with scalar loads)
template <int N, int CL = fm::kC_None> __device__ __forceinline__ double horner_tab(const double *tab, double x)
{
    return fm::horner_coefs<N, CL>(tab, x);
}
template <int N, int CL = fm::kC_None> __device__ __forceinline__ float horner_tab(const double *tab, float x)
{
    float p = (float)tab[N - 1];
#pragma unroll
    for (int i = N - 2; i >= 0; --i) p = __builtin_fmaf(p, x, (float)tab[i]);
    return p;
}
// ---- piecewise tables in LDS (kernels that define AB_PSI_LDS_TABLES before including this header and call psi_tables_fill())
// psi_m of Kansas / Paulson in s = LOG(y) and COARE's convective psi in L = LOG(y) (the functions of the two sections below) on 28
// equal intervals of [0, 6.6875) / [0, 7.4453125), and e_sat(T) of Goff (mod_phymbl.f90:777-800, in Pa) on 24 intervals of
// [265, 312) K; degree 7 on each interval (tools/gen_psitab.py; 1.7e-16 / 3.0e-16 absolute, 8.8e-17 relative: the rounding of
// the coefficients): 5 KB of LDS per block, coefficient-major, so that lanes on different intervals hit different banks.  An
// evaluation is five operations for the interval and the local variable, a shift for the address and seven FMAs, the eight
// coefficients arriving over the LDS pipe — where the global polynomials take 22 and 24 FMAs, and e_sat 14 FMAs plus a
// table-driven exponential, on the VALU, the unit that binds these kernels.  A timing experiment with polynomials of that
// length bounded the gain at 5 % (COARE + skin) and 9 % (no skin) before the psi tables were built (profiles/r2_notes.md).
// psi_m (Kansas / Paulson) in s = LOG(y): 28 intervals, max |table - function| = 1.74e-16
AB_TAB double kPsiTabM[224] = {
    0.03052775705881332, 0.09569459816076224, 0.16649173498048575, 0.24306809907978588,
    0.3255373854005741, 0.41397700911275687, 0.5084280413731213, 0.6088960675764189,
    0.7153528659332404, 0.8277387704664977, 0.9459655623341142, 1.0699197265915636,
    1.199465916528804, 1.3344504820390828, 1.474704939159095, 1.6200492820379016,
    1.7702950635759997, 1.9252481948058313, 2.0847114343599187, 2.248486557282549,
    2.416376206674841, 2.5881854422897566, 2.7637230075374064, 2.942802340891784,
    3.125242359935898, 3.3108680467660823, 3.499510862675603, 3.6910090183626405,
    0.031204886571636177, 0.03397715791512851, 0.036832406802617554, 0.03975344056493264,
    0.042722297486414575, 0.04572074154600107, 0.04873074201122734, 0.05173491249611595,
    0.054716889405301915, 0.05766163622829874, 0.06055566701898868, 0.06338718879595336,
    0.06614616793518777, 0.06882432955751876, 0.07141510134401277, 0.07391351424823855,
    0.07631607245161438, 0.07862060392904273, 0.08082610146582168, 0.08293256216971825,
    0.08494083167500442, 0.08685245749632754, 0.08866955445822267, 0.09039468384977874,
    0.09203074694468139, 0.0935808927702025, 0.09504843947513002, 0.09643680829950728,
    0.0006813353371241145, 0.0007041342259125406, 0.0007227718336838751, 0.0007369945639184909,
    0.000746672048977916, 0.0007517968588122264, 0.0007524773561120927, 0.0007489247980486743,
    0.0007414362040625, 0.0007303747028428929, 0.0007161490471518711, 0.0006991937831634975,
    0.0006799512414631592, 0.0006588561442141945, 0.0006363232541051578, 0.0006127381674278633,
    0.0005884511004364536, 0.000563773343375515, 0.000538975956286597, 0.0005142902431643387,
    0.0004899095509211493, 0.0004659919813074102, 0.00044266366359377784, 0.000420022302478213,
    0.0003981407814680827, 0.0003770706618950755, 0.0003568454690670414, 0.00033748369889548795,
    4.127559437475423e-06, 3.4617589477758207e-06, 2.743712946799944e-06, 1.993576819526483e-06,
    1.2320654398309443e-06, 4.792510265219546e-07, -2.465020544196497e-07, -9.293781529820021e-07,
    -1.5566754334010277e-06, -2.119081844731723e-06, -2.6106823187424425e-06, -3.0287424598339737e-06,
    -3.373328147038388e-06, -3.646824240565006e-06, -3.853410995109635e-06, -3.998546657960362e-06,
    -4.088491952195642e-06, -4.129899121654917e-06, -4.129476625381295e-06, -4.093731341084459e-06,
    -4.028783562011731e-06, -3.940245993789422e-06, -3.833155963597368e-06, -3.7119496282786124e-06,
    -3.58046760324456e-06, -3.4419826939307842e-06, -3.299241957409948e-06, -3.1545169144142243e-06,
    -7.916829140090213e-08, -8.689671698407174e-08, -9.219227462026078e-08, -9.490764404626353e-08,
    -9.504977154847372e-08, -9.276946379724926e-08, -8.833812611829906e-08, -8.211570533455604e-08,
    -7.451461269761672e-08, -6.596423097625129e-08, -5.687973715599592e-08, -4.763769665527144e-08,
    -3.855952122461422e-08, -2.990268539460097e-08, -2.1858719605941388e-08, -1.4556491958246473e-08,
    -8.069122884601244e-09, -2.4229675828273926e-09, 2.392646854249734e-09, 6.4159179119852594e-09,
    9.703956869326108e-09, 1.232544002487095e-08, 1.4354662176143612e-08, 1.5866969250432907e-08,
    1.6935443517756067e-08, 1.762866003805171e-08, 1.8009314181926494e-08, 1.8133525350457043e-08,
    -8.857639094562493e-10, -6.54942137011601e-10, -4.017059425725472e-10, -1.4150162827072004e-10,
    1.1052300799825258e-10, 3.409606519048898e-10, 5.392466674788325e-10, 6.983040569529862e-10,
    8.147056706982533e-10, 8.884102294150357e-10, 9.22184981039305e-10, 9.208519024157707e-10,
    8.904889726508847e-10, 8.37691879380982e-10, 7.689653510631449e-10, 6.902767531121146e-10,
    6.067742316049553e-10, 5.226506654690567e-10, 4.411232476855104e-10, 3.644950667565402e-10,
    2.942672080713355e-10, 2.312752310301979e-10, 1.758304310835484e-10, 1.2785270305552333e-10,
    8.698730298180299e-11, 5.2702039570999284e-11, 2.436440193618319e-11, 1.3000101952937403e-12,
    1.7882670004025236e-11, 2.0367792886558298e-11, 2.1598488652772115e-11, 2.153394244682336e-11,
    2.0262456842161683e-11, 1.7980471034525226e-11, 1.4958163374716286e-11, 1.1499088978008626e-11,
    7.901368500373368e-12, 4.426341804705842e-12, 1.2780527689590527e-12, -1.4057402782842982e-12,
    -3.55248090780079e-12, -5.14639585834831e-12, -6.215880692926163e-12, -6.819541109148499e-12,
    -7.033050434661618e-12, -6.9381642349897165e-12, -6.6144837900130434e-12, -6.133989462793558e-12,
    -5.558002546094944e-12, -4.936058147735612e-12, -4.306135245760123e-12, -3.695741728049886e-12,
    -3.123447840100227e-12, -2.600568955789507e-12, -2.1327981902626267e-12, -1.7216713047944073e-12,
    2.1898532076333346e-13, 1.3401002027023471e-13, 4.1332215565528644e-14, -4.9500971570289244e-14,
    -1.2981468832580349e-13, -1.9299410000071784e-13, -2.3518921321774244e-13, -2.5546184588991767e-13,
    -2.5543660008072107e-13, -2.3861410677537513e-13, -2.0954232124773003e-13, -1.7302405789650912e-13,
    -1.3348537584474725e-13, -9.456373372731918e-14, -5.891675223579186e-14, -2.8213038951662673e-14,
    -3.2481969585605714e-15, 1.5871856937263337e-14, 2.9528178190219994e-14, 3.8399990827391315e-14,
    4.3308552170185995e-14, 4.510064268397301e-14, 4.457054076602444e-14, 4.24147581433846e-14,
    3.921197389904554e-14, 3.542059996000773e-14, 3.1387425990005986e-14, 2.7362220958820366e-14};
// COARE convective psi in L = LOG(y): 28 intervals, max |table - function| = 2.95e-16
AB_TAB double kPsiTabC[224] = {
    0.04529948098209365, 0.14189234846792753, 0.24661017567867455, 0.35955352298642296,
    0.48075504594608287, 0.6101808768504351, 0.747734089344556, 0.8932599224377807,
    1.0465523549720699, 1.2073615824151538, 1.3754019521211005, 1.5503599530424756,
    1.7319019204183703, 1.9196811936716718, 2.113344546276124, 2.3125377816626242,
    2.5169104540305747, 2.726119724530255, 2.9398334011343463, 3.15773223557881,
    3.3795115648810437, 3.604882390379675, 3.833571986244687, 4.0653241240197735,
    4.299898991684401, 4.53707287629091, 4.776637669416922, 5.0184002451421135,
    0.046290391133633264, 0.050316467242499566, 0.05440974743396389, 0.05853630941680868,
    0.062662361968251, 0.06675534771018629, 0.07078491270516744, 0.07472368911604557,
    0.07854785976082436, 0.08223749630746098, 0.0857766829600353, 0.0891534527008305,
    0.09235957258748725, 0.09539021844313389, 0.09824357845570655, 0.10092042103931513,
    0.10342365618176914, 0.10575791261369409, 0.10792914640099159, 0.1099442905799116,
    0.11181095053319257, 0.113537146035445, 0.11513109821496434, 0.11660105794022725,
    0.1179551711590232, 0.11920137631095867, 0.12034732893257372, 0.12140034883900841,
    0.0009953416310716694, 0.0010163212118455532, 0.0010289005559759726, 0.0010329678291042393,
    0.0010286978168085042, 0.001016527027106686, 0.0009971127869080388, 0.000971281634366703,
    0.0009399728422009144, 0.000904182560910266, 0.0008649130587175914, 0.0008231301517304475,
    0.0007797304571360983, 0.0007355187965635099, 0.000691195067971766, 0.0006473492443284129,
    0.0006044628271567311, 0.0005629150207524989, 0.0005229920191157457, 0.0004848980349650175,
    0.00044876698445599217, 0.0004146740260199319, 0.0003826464080686944, 0.00035267329357035137,
    0.0003247143955267429, 0.0002987073787023812, 0.00027457406568436716, 0.00025222553700215224,
    4.178173373109639e-06, 2.803765721441123e-06, 1.3862571800150156e-06, -2.5573579325999376e-08,
    -1.3855228409764024e-06, -2.653002248822162e-06, -3.7952926549550493e-06, -4.78884365811298e-06,
    -5.619607318037561e-06, -6.282526775530032e-06, -6.780387991100389e-06, -7.122279164248291e-06,
    -7.321893631907094e-06, -7.395871737718539e-06, -7.362321059697416e-06, -7.239596241552893e-06,
    -7.045369040101725e-06, -6.795980995001981e-06, -6.506046415283076e-06, -6.188260674115348e-06,
    -5.853365355733584e-06, -5.510224575967214e-06, -5.165973105697561e-06, -4.826204686894181e-06,
    -4.495176730529953e-06, -4.176014591556879e-06, -3.8709044366193895e-06, -3.5812682627329217e-06,
    -1.6708124301857214e-07, -1.755163519602266e-07, -1.7783707608487502e-07, -1.741515703260154e-07,
    -1.6498538216026096e-07, -1.512000962759504e-07, -1.3388121118123843e-07, -1.1421501031742399e-07,
    -9.337303430561779e-08, -7.241817033805636e-08, -5.2240085522406406e-08, -3.3521501002397186e-08,
    -1.673191363659604e-08, -2.1424187137801872e-09, 1.0145754555907471e-08, 2.016594250043713e-08,
    2.804959128971167e-08, 3.399256836955514e-08, 3.822653040840922e-08, 4.0996293585630496e-08,
    4.254318667536688e-08, 4.309376878499703e-08, 4.285299628007124e-08, 4.200084352049047e-08,
    4.0691439922038166e-08, 3.905391721177789e-08, 3.7194319558809256e-08, 3.519808628048995e-08,
    -1.1415332950006071e-09, -5.397285986110223e-10, 7.372219848236827e-11, 6.544804842613599e-10,
    1.164206108691646e-09, 1.5745501550272189e-09, 1.8692600045982584e-09, 2.0442965306444627e-09,
    2.1062762435420714e-09, 2.0698135827904745e-09, 1.9544158444981247e-09, 1.7815025716811025e-09,
    1.5719457750487915e-09, 1.3443260909518624e-09, 1.1139257064133312e-09, 8.923591921524775e-10,
    6.876828340453735e-10, 5.04811129474378e-10, 3.460890649614219e-10, 2.1190424759955593e-10,
    1.0126147366414525e-10, 1.2276280138662427e-11, -5.7429966439500293e-11, -1.1043332422072647e-10,
    -1.492984914814855e-10, -1.7644285780776798e-10, -1.940572351150782e-10, -2.040682817612549e-10,
    4.836257811225434e-11, 5.123785237365566e-11, 5.0314634383715417e-11, 4.5883241032265463e-11,
    3.863218884233346e-11, 2.9508472734920564e-11, 1.9549990385060043e-11, 9.72830074285838e-12,
    8.307714760785015e-13, -6.603731669435805e-12, -1.2296428604148809e-11, -1.619909490662978e-11,
    -1.8439955934519472e-11, -1.9261187172549524e-11, -1.8960464403963515e-11, -1.7844010344580736e-11,
    -1.6193957232734443e-11, -1.4249460096520055e-11, -1.2199057418830645e-11, -1.0181065663113647e-11,
    -8.288918549044041e-12, -6.578930268935796e-12, -5.078674381512045e-12, -3.794840206532353e-12,
    -2.719966234480612e-12, -1.8378322659166164e-12, -1.1275334275684914e-12, -5.663906051781827e-13,
    3.3962374534011216e-13, 6.898763685989277e-14, -1.9740838612081193e-13, -4.2747345702643247e-13,
    -5.971795865833127e-13, -6.93954477141925e-13, -7.172524390048274e-13, -6.766195955596587e-13,
    -5.881820728387847e-13, -4.706881351354241e-13, -3.420610913229442e-13, -2.1702933343091417e-13,
    -1.0597886679959866e-13, -1.4855570258409276e-14, 5.4218566173759395e-14, 1.0197411783823857e-13,
    1.309599445703994e-13, 1.4463530581027375e-13, 1.4667873046509878e-13, 1.405295272032449e-13,
    1.291329176121607e-13, 1.1484012642507802e-13, 9.941244862721327e-14, 8.40855540677208e-14,
    6.966118885884547e-14, 5.660431709407978e-14, 4.5132796034851175e-14, 3.5293328263782466e-14};
// e_sat(T) [Pa] on 265 K <= T < 312 K, RELATIVE error: 24 intervals, max |table - function| = 7.82e-17
AB_TAB double kEsatTab[192] = {
    357.24731931304785, 414.99422650940187, 480.8987996855304, 555.9404053510857,
    641.1937569443639, 737.8357614239696, 847.1526368401537, 970.5472954120983,
    1109.5469855100807, 1265.8111848316862, 1441.139735976544, 1637.4812145717033,
    1856.9415190867292, 2101.792670510538, 2374.4818091471484, 2677.6403749307483,
    3014.0934568670914, 3386.8692964830975, 3799.2089295139494, 4254.5759494806625,
    4756.6663763142915, 5309.418612768144, 5917.02347102863, 6583.934251690103,
    26.98600824636144, 30.83506241256424, 35.15106987493433, 39.98003818371891,
    45.37132966983028, 51.37779811379038, 58.05592296162294, 65.465940520899,
    73.67197157702789, 82.74214488056312, 92.74871597066502, 103.76818081780274,
    115.88138379013232, 129.17361947257112, 143.7347278951979, 159.65918275801175,
    177.0461722720388, 195.99967227201776, 216.62851129315428, 239.04642734342136,
    263.3721161433108, 289.72927064651407, 318.24661169743405, 349.05790972440803,
    0.9075255835656063, 1.0187838881623978, 1.1411362466431239, 1.2754041017436184,
    1.4224433225964495, 1.58314366265069, 1.7584280753524943, 1.9492518888283756,
    2.1566018417699597, 2.3814949836560637, 2.6249774433571655, 2.8881230710411576,
    3.1720319591307495, 3.4778288488456406, 3.8066614295909744, 4.159698539122614,
    4.538128273025212, 4.9431560125773375, 5.376002380546369, 5.837901134852318,
    6.33009701036302, 6.853843519332522, 7.410400721170095, 8.001032972329853,
    0.01766273344130855, 0.019445219272153265, 0.021361719184827575, 0.023418008163795745,
    0.02561978272878078, 0.027972637178446138, 0.030482039960569127, 0.033153310329371376,
    0.03599159544806593, 0.03900184809069535, 0.042188805092049704, 0.0455569666879576,
    0.049110576880636594, 0.052853604955180045, 0.05678972826376008, 0.06092231638385929,
    0.06525441674592963, 0.06978874181443905, 0.07452765789442768, 0.07947317562357832,
    0.084626942197529, 0.08999023536383395, 0.09556395920772352, 0.1013486417407271,
    0.00021467733159965402, 0.00023106562811270372, 0.00024817971197498457, 0.00026601126007818406,
    0.00028454897860844897, 0.0003037786087524405, 0.00032368295258987503, 0.00034424191893583814,
    0.0003654325887194268, 0.00038722929931755463, 0.00040960374710559606, 0.0004325251073412866,
    0.00045596017036606527, 0.00047987349298974334, 0.0005042275638207023, 0.0005289829812152414,
    0.0005540986424464864, 0.000579531942635511, 0.0006052389819449231, 0.0006311747795078377,
    0.0006572934925524949, 0.000683548639184202, 0.000709893323301111, 0.000736280460147787,
    1.602336891035801e-06, 1.6752330203033494e-06, 1.7474450578751469e-06, 1.8186763230346087e-06,
    1.8886296930256147e-06, 1.9570096274549308e-06, 2.0235241781173077e-06, 2.087886966176499e-06,
    2.149819109452009e-06, 2.2090510835434253e-06, 2.265324501650969e-06, 2.318393799200852e-06,
    2.3680278107354905e-06, 2.414011227959733e-06, 2.4561459293236713e-06, 2.4942521730494816e-06,
    2.5281696470542032e-06, 2.557758370763518e-06, 2.582899445335916e-06, 2.603495650305911e-06,
    2.6194718860945483e-06, 2.6307754632123066e-06, 2.6373762402821375e-06, 2.6392666142291517e-06,
    6.094765446410906e-09, 6.050062482498653e-09, 5.980712268697602e-09, 5.88659361094141e-09,
    5.767754092812704e-09, 5.624409632847222e-09, 5.456942507409454e-09, 5.265897899720519e-09,
    5.051979050248029e-09, 4.816041096603777e-09, 4.5590837022653754e-09, 4.282242582801336e-09,
    3.986780045824054e-09, 3.6740746666373787e-09, 3.3456102255256808e-09, 3.002964034911973e-09,
    2.64779478527563e-09, 2.2818300378636038e-09, 1.906853489964331e-09, 1.5246921349629112e-09,
    1.137203434689662e-09, 7.462626158472489e-10, 3.5375019569248907e-10, -3.8460165203316975e-11,
    -2.318644221318079e-12, -4.070853390546024e-12, -5.837729791075424e-12, -7.607239227392822e-12,
    -9.367323170999578e-12, -1.110601015134346e-11, -1.2811523380341088e-11, -1.447238350740212e-11,
    -1.6077505519413996e-11, -1.761628892406362e-11, -1.907870048283806e-11, -2.0455348889867745e-11,
    -2.1737550922415065e-11, -2.2917388716405914e-11, -2.3987757944278766e-11, -2.494240679111141e-11,
    -2.5775965737211742e-11, -2.6483968260063447e-11, -2.706286266484239e-11, -2.7510015340004354e-11,
    -2.7823705812219242e-11, -2.8003114042898803e-11, -2.804830046660788e-11, -2.7960179319789032e-11};
constexpr double kPsiTabSMax = 6.6875, kPsiTabLMax = 7.4453125, kEsatTabT0 = 265., kEsatTabT1 = 312.;
constexpr int kTabPsikM = 0, kTabPsic = 1, kTabEsat = 2;
constexpr int kTabNint[3] = {28, 28, 24}, kTabOff[3] = {0, 224, 448}, kTabTotal = 640;
#ifdef AB_PSI_LDS_TABLES
constexpr bool kPsiTabDefault = true;
#else
constexpr bool kPsiTabDefault = false;
#endif
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
// two arrays: a kernel that never evaluates e_sat through its table (ESAT = false below) does not allocate the second
static __shared__ double s_psitab[kTabOff[1]];     // Kansas psi_m (the fp64 kernels of ECMWF, ANDREAS; COARE reads its psi through L1, ab_gtables.hpp)
static __shared__ double s_psitabc[kTabOff[2] - kTabOff[1]];   // convective psi (unused by the flux kernels since round 3; math_test_kernel)
static __shared__ double s_esattab[kTabTotal - kTabOff[2]];
// all threads of a block, BEFORE the barrier of math_tables_init().  ESAT: also the e_sat table (the kernels with the skin schemes,
// where e_sat is evaluated a dozen times per cell; without them four times: not worth 1.5 KB of the LDS five blocks per CU share)
template <bool ESAT = true> __device__ __forceinline__ void psi_tables_fill()
{
    for (int t = (int)threadIdx.x; t < kTabOff[1]; t += (int)blockDim.x) s_psitab[t] = kPsiTabM[t];
    if constexpr (ESAT)
        for (int t = (int)threadIdx.x; t < kTabTotal - kTabOff[2]; t += (int)blockDim.x) s_esattab[t] = kEsatTab[t];
}
// the e_sat table alone (the mixed kernels: fp32 psi tables, fp64 q_sat); all threads, before the barrier of math_tables_init()
__device__ __forceinline__ void esat_table_fill()
{
    for (int t = (int)threadIdx.x; t < kTabTotal - kTabOff[2]; t += (int)blockDim.x) s_esattab[t] = kEsatTab[t];
}
template <int WHICH> __device__ __forceinline__ double psi_tab_coef(int k, int i)
{
    if constexpr (WHICH == kTabEsat) return s_esattab[k * kTabNint[WHICH] + i];
    else if constexpr (WHICH == kTabPsic) return s_psitabc[k * kTabNint[WHICH] + i];
    else return s_psitab[k * kTabNint[WHICH] + i];
}
#else
template <int WHICH> inline double psi_tab_coef(int k, int i)
{
    const double *tab = WHICH == kTabPsikM ? &kPsiTabM[0] : (WHICH == kTabPsic ? &kPsiTabC[0] : &kEsatTab[0]);
    return tab[k * kTabNint[WHICH] + i];
}
#endif
// position in a piecewise table: interval = trunc(xn), local variable = 2 fract(xn) - 1 (xn >= 0): v_cvt_i32_f64, v_fract_f64 and one FMA
// (floor, subtract, FMA and the conversion were four)
__device__ __forceinline__ double tab_fract(double xn)
{
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
    return __builtin_amdgcn_fract(xn);
#else
    return xn - __builtin_floor(xn);
#endif
}
// xn = position in units of intervals, 0 <= xn < kTabNint[WHICH]
template <int WHICH> __device__ __forceinline__ double psi_tab_eval(double xn)
{
    AB_COUNT(WHICH == kTabPsikM ? "tab_psik_m" : (WHICH == kTabPsic ? "tab_psic" : "tab_e_sat"), 12.);   // floor, sub, fma, cvt, shift + 7 FMAs; the coefficients arrive over the LDS pipe
    const int i = (int)xn;
    const double u = __builtin_fma(tab_fract(xn), 2., -1.);
    double p = psi_tab_coef<WHICH>(7, i);
#pragma unroll
    for (int k = 6; k >= 0; --k) p = __builtin_fma(p, u, psi_tab_coef<WHICH>(k, i));
    return p;
}
// fp32: psi_m, psi_h (Kansas / Paulson), convective psi; 32 intervals, degree 3, [function][coefficient][interval]; max |table - function| = 7.0e-09 / 9.7e-09 / 1.3e-08
AB_TAB float kPsiTab32[384] = {
    0.0266377889f, 0.0830520019f, 0.143753141f, 0.208846718f, 0.278417915f, 0.352530777f,
    0.431228042f, 0.514531136f, 0.602440894f, 0.694938362f, 0.791986048f, 0.893529356f,
    0.999498308f, 1.10980904f, 1.22436619f, 1.34306383f, 1.46578801f, 1.59241807f,
    1.72282839f, 1.85688961f, 1.99447012f, 2.13543725f, 2.27965832f, 2.42700148f,
    2.57733607f, 2.73053432f, 2.88647056f, 3.04502273f, 3.20607162f, 3.36950254f,
    3.53520393f, 3.70306897f, 0.027155403f, 0.0292692184f, 0.0314407162f, 0.0336599648f,
    0.0359165668f, 0.0381998643f, 0.0404991768f, 0.0428039916f, 0.0451041833f, 0.047390148f,
    0.0496529676f, 0.0518844947f, 0.0540774353f, 0.0562253892f, 0.0583228618f, 0.0603652634f,
    0.0623488612f, 0.0642707422f, 0.0661287606f, 0.0679214597f, 0.0696480051f, 0.0713081136f,
    0.0729019865f, 0.0744302273f, 0.0758938044f, 0.0772939473f, 0.0786321312f, 0.0799100175f,
    0.0811294019f, 0.0822921842f, 0.0834003463f, 0.0844558999f, 0.00052041054f, 0.000536015024f,
    0.000549215591f, 0.000559866778f, 0.00056787784f, 0.000573213736f, 0.0005758935f, 0.000575986807f,
    0.000573608093f, 0.000568909862f, 0.000562074885f, 0.000553307997f, 0.000542828464f, 0.000530862308f,
    0.000517635955f, 0.000503370364f, 0.000488276652f, 0.000472552725f, 0.0004563806f, 0.000439925032f,
    0.000423332822f, 0.000406732695f, 0.000390235859f, 0.000373936782f, 0.000357914425f, 0.000342233398f,
    0.000326945534f, 0.000312091201f, 0.000297700753f, 0.000283795758f, 0.000270390301f, 0.000257492182f,
    2.79110645e-06f, 2.40543864e-06f, 1.99110832e-06f, 1.55702253e-06f, 1.11245993e-06f, 6.66647907e-07f,
    2.28367412e-07f, -1.9438788e-07f, -5.94678795e-07f, -9.66802418e-07f, -1.30638648e-06f, -1.61040248e-06f,
    -1.87710737e-06f, -2.10593134e-06f, -2.29732541e-06f, -2.45258843e-06f, -2.57368515e-06f, -2.66306938e-06f,
    -2.72352031e-06f, -2.75799857e-06f, -2.76952278e-06f, -2.76107289e-06f, -2.73551382e-06f, -2.69554039e-06f,
    -2.64364076e-06f, -2.58207592e-06f, -2.51287065e-06f, -2.43781437e-06f, -2.35846937e-06f, -2.2761833e-06f,
    -2.19210551e-06f, -2.10720577e-06f, 0.0529284403f, 0.16287373f, 0.278242528f, 0.398976237f,
    0.524988532f, 0.65616715f, 0.79237622f, 0.933458984f, 1.07924044f, 1.22953069f,
    1.38412833f, 1.54282284f, 1.70539832f, 1.87163615f, 2.04131699f, 2.21422386f,
    2.39014363f, 2.56886864f, 2.7501986f, 2.93394041f, 3.11991096f, 3.30793571f,
    3.49784994f, 3.68949938f, 3.88273907f, 4.07743406f, 4.27345848f, 4.47069693f,
    4.66904116f, 4.86839247f, 5.06865931f, 5.2697587f, 0.0536106117f, 0.0563322119f,
    0.059031684f, 0.0616948009f, 0.0643081069f, 0.0668591782f, 0.0693368316f, 0.0717312917f,
    0.0740343034f, 0.0762391835f, 0.0783408359f, 0.0803356841f, 0.0822216198f, 0.0839978829f,
    0.0856649131f, 0.0872242227f, 0.0886782259f, 0.0900300965f, 0.0912836194f, 0.0924430266f,
    0.0935129076f, 0.0944980606f, 0.0954034105f, 0.0962339118f, 0.0969944894f, 0.0976899713f,
    0.0983250439f, 0.0989042148f, 0.0994317904f, 0.0999118686f, 0.100348294f, 0.100744702f,
    0.000681870733f, 0.000678163779f, 0.000670830079f, 0.00066002633f, 0.000645978784f, 0.000628973241f,
    0.000609343173f, 0.000587456394f, 0.000563701382f, 0.000538474007f, 0.000512165076f, 0.000485149998f,
    0.000457779883f, 0.000430374843f, 0.00040321963f, 0.000376560987f, 0.000350606831f, 0.000325526897f,
    0.000301454653f, 0.000278489752f, 0.000256701489f, 0.000236132168f, 0.00021680085f, 0.000198706926f,
    0.000181833602f, 0.000166151018f, 0.000151619155f, 0.000138190415f, 0.000125811741f, 0.00011442657f,
    0.000103976359f, 9.44018975e-05f, -3.10091991e-07f, -9.2353946e-07f, -1.5170865e-06f, -2.07847256e-06f,
    -2.59686453e-06f, -3.06327524e-06f, -3.47084733e-06f, -3.81499649e-06f, -4.09341146e-06f, -4.30592809e-06f,
    -4.45430442e-06f, -4.54192104e-06f, -4.57343867e-06f, -4.5544466e-06f, -4.49111849e-06f, -4.38990128e-06f,
    -4.25724784e-06f, -4.09939685e-06f, -3.92220591e-06f, -3.73103489e-06f, -3.53067253e-06f, -3.32530203e-06f,
    -3.11849635e-06f, -2.91323863e-06f, -2.71195813e-06f, -2.51657821e-06f, -2.32856996e-06f, -2.14900979e-06f,
    -1.97863346e-06f, -1.81789073e-06f, -1.66699499e-06f, -1.52596692e-06f, 0.0395287387f, 0.123166457f,
    0.213005647f, 0.309125036f, 0.411563545f, 0.520320535f, 0.635356963f, 0.756597757f,
    0.883934677f, 1.01722944f, 1.15631783f, 1.30101407f, 1.45111418f, 1.60640061f,
    1.76664603f, 1.93161643f, 2.10107493f, 2.27478409f, 2.45250845f, 2.63401699f,
    2.81908441f, 3.00749183f, 3.19902921f, 3.39349484f, 3.59069681f, 3.790452f,
    3.99258733f, 4.19693995f, 4.4033556f, 4.61168957f, 4.82180643f, 5.03357935f,
    0.0402865335f, 0.0433610156f, 0.0464847423f, 0.0496378988f, 0.0528005436f, 0.055953145f,
    0.059077017f, 0.0621547587f, 0.0651705638f, 0.068110466f, 0.0709624812f, 0.0737167001f,
    0.0763652474f, 0.0789022371f, 0.0813236162f, 0.0836270303f, 0.0858116299f, 0.0878778696f,
    0.0898273215f, 0.091662474f, 0.0933865607f, 0.0950033888f, 0.0965171903f, 0.0979325101f,
    0.0992540643f, 0.100486681f, 0.101635203f, 0.102704428f, 0.103699051f, 0.104623668f,
    0.105482683f, 0.106280334f, 0.00076074939f, 0.000775490596f, 0.000785338576f, 0.000790198392f,
    0.000790105667f, 0.000785219541f, 0.000775809516f, 0.000762237469f, 0.000744936522f, 0.00072438881f,
    0.000701103534f, 0.000675596471f, 0.000648372574f, 0.000619911356f, 0.000590656302f, 0.00056100724f,
    0.000531316269f, 0.000501885894f, 0.000472969958f, 0.000444775535f, 0.000417466654f, 0.000391168258f,
    0.000365970744f, 0.000341934705f, 0.000319095183f, 0.000297466147f, 0.000277044106f, 0.0002578118f,
    0.000239740984f, 0.000222795279f, 0.000206932193f, 0.000192105043f, 2.85429064e-06f, 2.05379683e-06f,
    1.22649419e-06f, 3.94353407e-07f, -4.2125194e-07f, -1.20067079e-06f, -1.92689845e-06f, -2.58623254e-06f,
    -3.16863543e-06f, -3.66780728e-06f, -4.08100732e-06f, -4.40867916e-06f, -4.65394305e-06f, -4.82202267e-06f,
    -4.91965784e-06f, -4.95455151e-06f, -4.93488278e-06f, -4.86889667e-06f, -4.76458672e-06f, -4.62946218e-06f,
    -4.47038792e-06f, -4.29349984e-06f, -4.10416487e-06f, -3.90698779e-06f, -3.70584439e-06f, -3.50393452e-06f,
    -3.30384864e-06f, -3.10763653e-06f, -2.91687957e-06f, -2.73275714e-06f, -2.55611189e-06f, -2.38750613e-06f};
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
static __shared__ float s_psitab32[384];
// fp32 kernels: all threads of a block of 256, before the block's first barrier
__device__ __forceinline__ void psi_tables_fill32()
{
    for (int t = (int)threadIdx.x; t < 384; t += (int)blockDim.x) s_psitab32[t] = kPsiTab32[t];
}
template <int WHICH> __device__ __forceinline__ float psi_tab_coef32(int k, int i) { return s_psitab32[WHICH * 128 + k * 32 + i]; }
#else
template <int WHICH> inline float psi_tab_coef32(int k, int i) { return kPsiTab32[WHICH * 128 + k * 32 + i]; }
#endif
// WHICH: 0 psi_m, 1 psi_h (Kansas / Paulson, x32 = 32 s / 6.6875), 2 convective psi (x32 = 32 L / 7.4453125); 0 <= x32 < 32
template <int WHICH> __device__ __forceinline__ float psi_tab_eval32(float x32)
{
    const float fi = __builtin_floorf(x32);
    const int i = (int)fi;
    const float u = __builtin_fmaf(x32 - fi, 2.f, -1.f);
    float p = psi_tab_coef32<WHICH>(3, i);
#pragma unroll
    for (int k = 2; k >= 0; --k) p = __builtin_fmaf(p, u, psi_tab_coef32<WHICH>(k, i));
    return p;
}
}  // namespace ab
#include "ab_gtables.hpp"
namespace ab {
// ---- piecewise tables read through the vector-memory path (ab_gtables.hpp, tools/gen_gtab.py).  Interval-major: one address, four
// 16-byte loads that hit L1 (measured equal to LDS for the e_sat table; LDS is full, the L1 is idle).  xn = position in units of
// intervals, 0 <= xn < number of intervals.
struct GtabPos {
    int i;
    double u;
};
__device__ __forceinline__ GtabPos gtab_pos(double xn)
{
    return GtabPos{(int)xn, __builtin_fma(tab_fract(xn), 2., -1.)};
}
__device__ __forceinline__ double gtab_at(const double *tab, const GtabPos &p)
{
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const d2 *q = (const d2 *)(tab + 8 * p.i);
    const d2 c01 = q[0], c23 = q[1], c45 = q[2], c67 = q[3];
    double r = c67[1];
    r = __builtin_fma(r, p.u, c67[0]); r = __builtin_fma(r, p.u, c45[1]); r = __builtin_fma(r, p.u, c45[0]);
    r = __builtin_fma(r, p.u, c23[1]); r = __builtin_fma(r, p.u, c23[0]); r = __builtin_fma(r, p.u, c01[1]);
    return __builtin_fma(r, p.u, c01[0]);
#else
    const double *c = tab + 8 * p.i;
    double r = c[7];
    for (int k = 6; k >= 0; --k) r = fm::p_fma(r, p.u, c[k]);
    return r;
#endif
}
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
static __shared__ double s_csgtab[8 * kCsGTabN];
// the cool skin's g(u) table in LDS (tiled fp64 COARE kernels with the skin schemes); all threads, before the barrier of math_tables_init()
__device__ __forceinline__ void csg_table_fill()
{
    for (int t = (int)threadIdx.x; t < 8 * kCsGTabN; t += (int)blockDim.x) s_csgtab[t] = kCsGTab[t];
}
__device__ __forceinline__ double csg_coef(int k, int i) { return s_csgtab[k * kCsGTabN + i]; }
#else
inline double csg_coef(int k, int i) { return kCsGTab[k * kCsGTabN + i]; }
#endif
// g(u) from the LDS table; xn = u * kCsGTabN / kCsGTabMax in [0, kCsGTabN)
__device__ __forceinline__ double csg_lds_eval(double xn)
{
    const int i = (int)xn;
    const double u = __builtin_fma(tab_fract(xn), 2., -1.);
    double p = csg_coef(7, i);
#pragma unroll
    for (int k = 6; k >= 0; --k) p = __builtin_fma(p, u, csg_coef(k, i));
    return p;
}
template <class R, bool TAB = kPsiTabDefault> __device__ __forceinline__ R e_sat(R pTa)
{
    AB_REGION("e_sat");
    using M = Mth<R>;
    const R zta = vmax(pTa, R(180.));
    if constexpr (sizeof(R) == 8 && TAB) {   // the tiled fp64 flux kernels: e_sat itself from its piecewise LDS table
        if ((zta >= R(kEsatTabT0)) && (zta < R(kEsatTabT1)))
            return R(psi_tab_eval<kTabEsat>((double)R((zta - R(kEsatTabT0)) * R(24. / (kEsatTabT1 - kEsatTabT0)))));
    }
    R e;
    if ((zta >= R(265.)) && (zta <= R(312.))) {
        e = goff_poly((zta - R(288.5)) * R(1. / 23.5));
    } else {
        const R ztmp = M::div(K<R>::rt0, zta);
        const R zx = zta * R(1. / 273.15);
        e = R(10.79574) * (R(1.) - ztmp) - R(5.028) * M::log10(zx)
            + R(1.50475E-4) * (R(1.) - M::exp10(R(-8.2969) * (zx - R(1.))))
            + R(0.42873E-3) * (M::exp10(R(4.76955) * (R(1.) - ztmp)) - R(1.)) + R(0.78614);
    }
    return R(100.) * M::exp10(e);
}
// q_sat_sclr :881-904
template <class R, bool TAB = kPsiTabDefault> __device__ __forceinline__ R q_sat(R pTa, R pslp)
{
    AB_REGION("q_sat");
    const R ze_s = e_sat<R, TAB>(pTa);
    return Mth<R>::div(K<R>::reps0 * ze_s, pslp - K<R>::one_m_reps0 * ze_s);
}
// q_air_rh :963-985
template <class R, bool TAB = kPsiTabDefault> __device__ __forceinline__ R q_air_rh(R prha, R pTa, R pslp)
{
    const R ze = R(0.01) * prha * e_sat<R, TAB>(pTa);
    return Mth<R>::div(ze * K<R>::reps0, vmax(pslp - K<R>::one_m_reps0 * ze, R(1.)));
}
// q_air_dp :990-1000
template <class R, bool TAB = kPsiTabDefault> __device__ __forceinline__ R q_air_dp(R da, R pslp)
{
    const R q = vmax(e_sat<R, TAB>(da), R(0.));
    return Mth<R>::div(q * K<R>::reps0, vmax(pslp - K<R>::one_m_reps0 * q, R(1.)));
}
// Theta_from_z_P0_T_q :343-375 = Pz_from_P0_tz_qz_sclr :283-318 (3 barometric iterations) + Poisson :189-200
// exp(x) for |x| <= 2.5e-3: the barometric exponent g M z/(R T) of a measurement height of 10 m or less (1.9e-3 at 200 K).  Degree-5
// Taylor polynomial, truncation x^6/720 < 4e-19: five FMAs instead of the table-driven exponential (16 slots), four times per cell.
template <class R> __device__ __forceinline__ R exp_tiny(R x)
{
    R p = R(1. / 120.);
    p = p * x + R(1. / 24.);
    p = p * x + R(1. / 6.);
    p = p * x + R(0.5);
    p = p * x + R(1.);
    return p * x + R(1.);
}
template <class R, bool TAB = kPsiTabDefault> __device__ __forceinline__ R theta_from_z_p0_t_q(R pz, R pslp, R pTa, R pqa)
{
    using M = Mth<R>;
    // e_sat(pTa) does not depend on the pressure iterate: evaluate once (the reference recomputes it)
    const R ze_s = e_sat<R, TAB>(pTa);
    const R c = M::div(-K<R>::grav * pz, K<R>::R_gas * pTa);
    const R zi = M::rcp(K<R>::reps0 * ze_s);
    // |c M| <= 9.8 * 0.029 * pz / (8.3145 * pTa): below 2.5e-3 for pz <= 10 m and pTa >= 137 K (pz is wave-uniform)
    const bool tiny = (pz <= R(10.)) && (pTa >= R(137.));
    R zpa = pslp;
    R zarg = R(0.);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const R zf = pqa * ((zpa - K<R>::one_m_reps0 * ze_s) * zi);   // q / q_sat(T, p)
        const R zxm = (R(1.) - zf) * K<R>::rmm_dryair + zf * K<R>::rmm_water;
        zarg = c * zxm;
        zpa = pslp * (tiny ? exp_tiny(zarg) : M::exp(zarg));
    }
    // T (P0/Pz)^kappa with P0/Pz = exp(-zarg): no log needed
    const R za = -K<R>::rpoiss_dry * zarg;
    return pTa * (tiny ? exp_tiny(za) : M::exp(za));
}
// virt_temp_sclr :247-269
template <class R> __device__ __forceinline__ R virt_temp(R t, R q) { return t * (R(1.) + K<R>::rctv0 * q); }
// rho_air_sclr :522-537
template <class R> __device__ __forceinline__ R rho_air(R pTa, R pqa, R pslp)
{
    return vmax(Mth<R>::div(pslp, K<R>::R_dry * pTa * (R(1.) + K<R>::rctv0 * pqa)), R(0.8));
}
// visc_air_sclr :549-563
template <class R> __device__ __forceinline__ R visc_air(R pTa)
{
    const R ztc = pTa - K<R>::rt0;
    const R ztc2 = ztc * ztc;
    return R(1.326e-5) * (R(1.) + R(6.542E-3) * ztc + R(8.301e-6) * ztc2 - R(4.84e-9) * ztc2 * ztc);
}
// One_on_L_sclr :666-693
template <class R> __device__ __forceinline__ R one_on_l(R pThta, R pqa, R pus, R pts, R pqs)
{
    AB_REGION("one_on_l");
    const R zqa = R(1.) + K<R>::rctv0 * pqa;
    const R r = Mth<R>::div(K<R>::grav * K<R>::vkarmn * (pts * zqa + K<R>::rctv0 * pThta * pqs),
                            vmax(pus * pus * pThta * zqa, R(1.E-9)));
    return sclamp(r, R(200.));
}
// Ri_bulk_sclr :712-747 (layer arguments are never passed on this path)
// A: anchor type of the temperatures and humidities (A = R, or double with R = float in the mixed mode: the difference of the two
// virtual temperatures is formed from the anchors, everything after it in R — see "anchors" at CellIn)
template <class R, class A = R> __device__ __forceinline__ R ri_bulk(R pz, A psst, A pThta, A pssq, A pqa, R pub)
{
    AB_REGION("ri_bulk");
    const A zsstv = virt_temp(psst, pssq);
    const R zdthv = R(virt_temp(pThta, pqa) - zsstv);
    const R ztv = R(0.5) * (R(zsstv) + virt_temp(R(pThta) - K<R>::rgamma_dry * pz, R(pqa)));
    return Mth<R>::div(K<R>::grav * zdthv * pz, ztv * pub * pub);
}
// BULK_FORMULA_SCLR :1149-1203 (open ocean: l_ice false)
template <class R, class A = R>
__device__ __forceinline__ void bulk_formula(R pzu, A pts, A pqs, A pThta, A pqa, R pCd, R pCh, R pCe, R pwnd,
                                             R pUb, R pslp, R &pTau, R &pQsen, R &pQlat, R &pEvap)
{
    const R zta = R(pThta) - K<R>::rgamma_dry * pzu;
    // rho_air twice (:1183-1184) with the same denominator: one reciprocal
    const R zir = Mth<R>::rcp(K<R>::R_dry * zta * (R(1.) + K<R>::rctv0 * R(pqa)));
    R zrho = vmax(pslp * zir, R(0.8));
    zrho = vmax((pslp - zrho * K<R>::grav * pzu) * zir, R(0.8));
    const R zUrho = pUb * vmax(zrho, R(1.));
    pTau = zUrho * pCd * pwnd;
    pEvap = zUrho * pCe * R(pqa - pqs);
    pQsen = zUrho * pCh * R(pThta - pts) * (K<R>::rCp_dry + K<R>::rCp_vap * R(pqa));
    pQlat = (R(2.501) - R(0.00237) * (R(pts) - K<R>::rt0)) * R(1.e6) * pEvap;  // L_vap :590
}
// UPDATE_QNSOL_TAU_SCLR :1059-1103 (+ qlw_net_sclr :1291-1314).  The routine forms Ch = (u*/Ub) theta*/zdt and Ce = (u*/Ub) q*/zdq
// with zdt, zdq the floored air-sea differences (:1076-1077) and hands them to BULK_FORMULA, which multiplies them by the
// unfloored differences again (:1190-1191): Ch (theta - T_s) = (u*/Ub) theta* exactly unless |theta - T_s| < 1e-9 (likewise q
// with 1e-12).  The two divisions are therefore only executed on those (rare) lanes.
template <class R, class A = R>
__device__ __forceinline__ void update_qnsol_tau(R pzu, A pts_a, A pqs, A pThta_a, A pqa_a, R pust, R ptst, R pqst,
                                                 R pwnd, R pUb, R pslp, R prlw, R &pQns, R &pTau, R &pQlat)
{
    AB_REGION("update_qnsol_tau");
    using M = Mth<R>;
    const R dth = R(pThta_a - pts_a), dq = R(pqa_a - pqs);
    const R pts = R(pts_a), pThta = R(pThta_a), pqa = R(pqa_a);
    const R zz0 = M::div(pust, pUb);
    R chdt = zz0 * ptst, cedq = zz0 * pqst;                        // Ch (theta - T_s), Ce (q - q_s)
    if (M::abs(dth) < R(1.E-09)) chdt = M::div(chdt, sfloor(dth, R(1.E-09))) * dth;
    if (M::abs(dq) < R(1.E-12)) cedq = M::div(cedq, sfloor(dq, R(1.E-12))) * dq;
    // BULK_FORMULA_SCLR :1149-1203 with those products (see bulk_formula above)
    const R zta = pThta - K<R>::rgamma_dry * pzu;
    const R zir = M::rcp(K<R>::R_dry * zta * (R(1.) + K<R>::rctv0 * pqa));
    R zrho = vmax(pslp * zir, R(0.8));
    zrho = vmax((pslp - zrho * K<R>::grav * pzu) * zir, R(0.8));
    const R zUrho = pUb * vmax(zrho, R(1.));
    pTau = zUrho * (zz0 * zz0) * pwnd;
    const R zEvap = zUrho * cedq;
    const R zQsen = zUrho * chdt * (K<R>::rCp_dry + K<R>::rCp_vap * pqa);
    pQlat = (R(2.501) - R(0.00237) * (pts - K<R>::rt0)) * R(1.e6) * zEvap;  // L_vap :590
    const R zt2 = pts * pts;
    pQns = pQlat + zQsen + K<R>::emiss_w * (prlw - K<R>::stefan * zt2 * zt2);
}
// alpha_sw_sclr :1267-1280
template <class R> __device__ __forceinline__ R alpha_sw(R psst)
{
    AB_REGION("alpha_sw");
    return R(2.1e-5) * pow_pos(vmax(psst - K<R>::rt0 + R(3.2), R(0.)), R(0.79));
}

// ---------------------------------------------------------------- cool skin
// CS_COARE mod_skin_coare.f90:48-93 (c0 = 0.137, latent-heat term) and CS_ECMWF
// mod_skin_ecmwf.f90:68-110 (c0 = 0.065) around delta_skin_layer_sclr mod_phymbl.f90:2010-2046.
// palpha = alpha_sw(SST) is hoisted by the caller.
// GLDS: g(u) of the absorption profile from its LDS table (the caller's kernel filled it: csg_table_fill)
template <class R, bool COARE, bool GLDS = false> __device__ __forceinline__ R cool_skin(R pQsw, R pQnsol, R pustar, R palpha, R pQlat)
{
    AB_REGION("cool_skin");
    using M = Mth<R>;
    const R c0 = COARE ? R(0.137) : R(0.065);
    // invariants of the five delta evaluations
    const R zusw = vmax(pustar, R(1.E-4)) * K<R>::sq_radrw;
    const R ziu = M::rcp(zusw);
    const R ziu2 = ziu * ziu;
    // delta_skin_layer_sclr mod_phymbl.f90:2030-2044 forms zQd = Qd + 0.026 MIN(Qlat,0) Cp_w/Lv/alpha and x = alpha rcst_cs/u*w^4 zQd;
    // here x = (rcst_cs/u*w^4) (alpha Qd + 0.026 MIN(Qlat,0) Cp_w/Lv): the same number without the division (alpha > 0: same sign)
    const R zB = K<R>::rcst_cs * (ziu2 * ziu2);
    const R ztmp = K<R>::rnu0_w * ziu;
    const R zql = COARE ? R(0.026 * 4190. / 2.46e+6) * vmin(pQlat, R(0.)) : R(0.);
    const R zdwarm = vmin(R(6.) * ztmp, R(0.007));
    const R ziz6 = zusw * R(1. / (6. * 1.e-6));                     // 1/(6 nu/u*w), K<R>::rnu0_w = 1e-6
    // delta; 1/delta for the absorption profile below is (1 + x^0.75)^(1/3) / (6 nu/u*w) = y rcbrt(y)^2 u*w/(6 nu): no division.  Where
    // the profile comes from its table (fp64 COARE) 1/delta is only wanted for a layer thicker than the table covers: formed then
    // (LAZY: y and the cube root are kept instead — in the rare warm branch a pair that gives 1/delta by the same formula —; elsewhere
    // the one number 1/delta is kept, which holds fewer registers)
    constexpr bool LAZY = sizeof(R) == 8 && COARE;
    R zy = R(1.), zrc = R(1.), zidelta = R(0.);
    auto delta = [&](R pQd) -> R {
        const R zQd = palpha * pQd + zql;                           // alpha (Qd + zql/alpha)
        if (nonneg(zQd)) {                                          // warming of the viscous layer (rare)
            if (LAZY) { zrc = R(1.); zy = M::rcp(zdwarm * ziz6); } else zidelta = M::rcp(zdwarm);
            return zdwarm;
        }
        // x^0.75 = x x^(-1/4), the fourth root from an fp32 seed and one cubic step.  The reference's floor of x at 0 becomes 1e-30
        // (fp32 range of the seed): below it x^0.75 < 6e-23 and 1 + x^0.75 is 1 either way
        const R x = vmax(zB * zQd, R(1.E-30));
        R y, rc;
        if constexpr (sizeof(R) == 8) {
            AB_COUNT("qskin_pair", 24.);
            double yy, rr;
            fm::qskin_pair((double)x, yy, rr);                      // both fp32 seeds from one conversion of x
            y = R(yy); rc = R(rr);
        } else {
            y = R(1.) + x * M::rqrt(x);
            rc = M::rcbrt(y);
        }
        if (LAZY) { zy = y; zrc = rc; } else zidelta = (y * rc) * (rc * ziz6);
        return R(6.) * rc * ztmp;                                   // 6 (1 + x^0.75)^(-1/3) nu/u*w
    };
    auto inv_delta = [&]() -> R { return LAZY ? (zy * zrc) * (zrc * ziz6) : zidelta; };
    R zQabs = pQnsol;
    R zdelta = delta(zQabs);
    // no insolation (night side): zQabs = pQnsol + zfr*0 never changes, the four sub-iterations reproduce zdelta exactly
    if (pQsw != R(0.))
#pragma unroll 1
    for (int jc = 0; jc < 4; ++jc) {
        R zabs;      // 6.6e-5/delta (1 - exp(-delta/8e-4))
#ifndef AB_NO_GTABLES
        if constexpr (sizeof(R) == 8 && COARE) {   // (ECMWF + skin sits at the 128-VGPR limit: eight more coefficients in flight would spill)
            // = 0.0825 g(u), g(u) = (1 - exp(-u))/u, u = delta/8e-4: g from a piecewise table — in LDS where the kernel has one (twenty
            // evaluations per cell in a dependent chain: the L1 round trip shows), else through L1 (ab_gtables.hpp) — instead of an
            // exponential and the reciprocal thickness
            const R zu8 = zdelta * R(1. / 8.E-4);
            if (zu8 < R(GLDS ? kCsGTabMax : kGCsGMax)) {
                AB_COUNT("gtab_cs_g", 11.5);
                double g;
                if constexpr (GLDS) g = csg_lds_eval((double)R(zu8 * R(kCsGTabN / kCsGTabMax)));
                else g = gtab_at(kGCsG, gtab_pos((double)R(zu8 * R(kGCsGN / kGCsGMax))));
                zabs = R(6.6E-5 / 8.E-4) * R(g);
            } else {
                zabs = (R(6.6E-5) * inv_delta()) * (R(1.) - M::exp(-zu8));
            }
        } else
#endif
            zabs = (R(6.6E-5) * inv_delta()) * (R(1.) - M::exp(zdelta * R(-1. / 8.E-4)));
        const R zfr = vmax(c0 + R(11.) * zdelta - zabs, R(0.01));
        zQabs = pQnsol + zfr * pQsw;
        zdelta = delta(zQabs);
    }
    return zQabs * zdelta * R(1. / 0.6);
}

// ---------------------------------------------------------------- warm layer, COARE 3.6
// WL_COARE mod_skin_coare.f90:97-250.  State st = {dT_wl, Hz_wl, Qnt_ac, Tau_ac}.
// Hoisted per cell: zcd1, zcd2 (depend on alpha_sw(SST) only), `dawn` (solar time 4h-6.5h).
template <class R> struct WlCoareCell {
    R zcd1, zcd2;
    bool dawn;
};
template <class R> __device__ __forceinline__ R wl_absorb(R zHwl)
{
    // 1 - (0.28*0.014*(1-e^{-H/0.014}) + 0.27*0.357*(1-e^{-H/0.357}) + 0.45*12.82*(1-e^{-H/12.82}))/H  :167-168
    // exp(x) for x < -40 rounds 1-exp(x) to exactly 1 in fp64 (and fp32): skip those evaluations.
    using M = Mth<R>;
    const R a1 = zHwl * R(1. / 0.014), a2 = zHwl * R(1. / 0.357);
    const R e1 = a1 > R(40.) ? R(1.) : R(1.) - M::exp(-a1);
    const R e2 = a2 > R(40.) ? R(1.) : R(1.) - M::exp(-a2);
    // the deepest band: H/12.82 is 0.008 ... 1.6 and 1 - exp(-x) cancels for small x.  fp64 has the digits to spare; fp32 (also the
    // mixed mode) would leave 6e-8/x = 1e-5 in the absorbed fraction of a shallow, strongly heated layer, i.e. in its dT_wl of 1-3 K:
    // below x = 0.35 the series x (1 - x/2 + x^2/6 - ...) to x^7 (truncation 3e-9 relative)
    R e3;
    if constexpr (sizeof(R) == 4) {
        const R x = zHwl * R(1. / 12.82);
        if (x < R(0.35)) {
            R p = R(-1. / 5040.);
            p = p * x + R(1. / 720.); p = p * x + R(-1. / 120.); p = p * x + R(1. / 24.); p = p * x + R(-1. / 6.); p = p * x + R(0.5);
            e3 = x - (p * x) * x;
        } else {
            e3 = R(1.) - M::exp(-x);
        }
    } else {
        e3 = R(1.) - M::exp(zHwl * R(-1. / 12.82));
    }
    return R(1.) - M::div(R(0.28 * 0.014) * e1 + R(0.27 * 0.357) * e2 + R(0.45 * 12.82) * e3, zHwl);
}
template <class R>
__device__ __forceinline__ void wl_coare(R (&st)[4], const WlCoareCell<R> &c, R pQsw, R pQnsol, R pTau, bool commit)
{
    AB_REGION("wl_coare");
    using M = Mth<R>;
    const R Hwl_max = R(20.);
    R zdTwl = st[0];
    R zHwl = vmax(vmin(st[1], Hwl_max), R(0.1));
    R zqac = st[2];
    R ztac = st[3];
    bool l_exit = c.dawn, l_destroy = c.dawn;  // dawn reset :159-163
    R zQabs = R(0.);
    if (!l_exit) {
        zQabs = wl_absorb(zHwl) * pQsw + pQnsol;                                   // :167-169
        if ((M::abs(zdTwl) < R(1.E-6)) && (zQabs <= R(0.))) l_exit = true;         // :171-176
    }
    if (!l_exit && (st[2] + zQabs * K<R>::rdt <= R(0.))) { l_exit = true; l_destroy = true; }  // :182-185
    if (!l_exit) {
        ztac = st[3] + vmax(R(.002), pTau) * K<R>::rdt;                            // :199
#pragma unroll 1
        for (int jl = 0; jl < 5; ++jl) {                                           // :204-211
            // pass 0 re-evaluates the absorption at the same zHwl as the test above (:167-169): reuse it
            if (jl > 0) zQabs = wl_absorb(zHwl) * pQsw + pQnsol;
            zqac = st[2] + zQabs * K<R>::rdt;
            if (zqac <= R(0.)) break;
            zHwl = vmax(vmin(Hwl_max, c.zcd1 * ztac * M::rsqrt_pos(zqac)), R(0.1));
        }
        if (zqac <= R(0.)) {
            l_destroy = true;
        } else {
            zdTwl = M::div(c.zcd2 * (zqac * M::sqrt_pos(zqac)), ztac);             // :220 (zqac > 0 here)
            if (!nonneg(K<R>::gdept1 - zHwl)) zdTwl = M::div(zdTwl * K<R>::gdept1, zHwl);  // :223-224
        }
    }
    if (l_destroy) { zdTwl = R(0.); zHwl = Hwl_max; zqac = R(0.); ztac = R(0.); }   // :229-235
    if (commit) { st[0] = zdTwl; st[1] = zHwl; st[2] = zqac; st[3] = ztac; }        // :239-248
}
// local solar time test of WL_COARE :146-163 -> true inside the dawn-reset window ]4h, 6.5h]
template <class R> __device__ __forceinline__ bool wl_coare_dawn(R plon, int isd)
{
    using M = Mth<R>;
    auto fmodulo = [](R a, R p) -> R { return a - M::floor(M::div(a, p)) * p; };
    R rlag = R(-1.) * fmodulo((R(360.) - fmodulo(plon, R(360.))) * R(1. / 15.), R(24.));
    rlag = R(-1.) * M::copysign(vmin(M::abs(rlag), M::abs(fmodulo(rlag, R(24.)))), rlag + R(12.));
    const int ilag = (int)(rlag * R(3600.));
    int isd_sol = (isd + ilag) % 86400;
    if (isd_sol < 0) isd_sol += 86400;
    const R rhr = (R)isd_sol * R(1. / 3600.);
    return (rhr > R(4.)) && (rhr <= R(6.5));
}

// ---------------------------------------------------------------- warm layer, ECMWF (Zeng & Beljaars 2005 / Takaya 2010)
// PHI mod_skin_ecmwf.f90:233-253: only its zeta >= 0 branch is reachable from WL_ECMWF (both stability parameters are >= 0)
// WL_ECMWF mod_skin_ecmwf.f90:113-230 (no Stokes drift).  Advances dT_wl on EVERY call (:228).
// The layer depth is fixed in this scheme (Hz_wl is only ever read, :136): what depends on it alone — the absorbed fraction of
// the solar flux with its three exponentials (:152) and 1/Hz_wl — is evaluated once per record by the caller (WlEcmwfCell),
// not once per Monin-Obukhov iteration.
template <class R> struct WlEcmwfCell {
    R zfr, zih;
};
template <class R> __device__ __forceinline__ WlEcmwfCell<R> wl_ecmwf_cell(R zHwl)
{
    using M = Mth<R>;
    WlEcmwfCell<R> c;
    c.zfr = R(1.) - R(0.28) * M::exp(R(-71.5) * zHwl) - R(0.27) * M::exp(R(-2.8) * zHwl) - R(0.45) * M::exp(R(-0.07) * zHwl);
    c.zih = M::rcp(zHwl);
    return c;
}
template <class R>
__device__ __forceinline__ void wl_ecmwf(R &dT_wl, R zHwl, const WlEcmwfCell<R> &wc, R pQsw, R pQnsol, R pustar, R zalpha)
{
    AB_REGION("wl_ecmwf");
    using M = Mth<R>;
    const R zRhoCp_w = K<R>::rho0_w * K<R>::rCp0_w;
    const R rNuwl0 = R(0.5);
    const R zih = wc.zih;
    const bool deep = !nonneg(K<R>::gdept1 - zHwl);       // warm layer deeper than the bulk-SST depth (always: 3 m vs 1 m)
    const R ztcorr = deep ? K<R>::gdept1 * zih : R(1.);
    const R zdTwl_b = vmax(deep ? dT_wl * zHwl : dT_wl, R(0.));   // dT_wl / ztcorr, gdept1 == 1
    const R zQabs = wc.zfr * pQsw + pQnsol;
    const R zusw = vmax(pustar, R(1.E-4)) * K<R>::sq_radrw;
    const R zusw2 = zusw * zusw;
    const R zfLa = R(2.231443166940565);  // MAX(0.3**(-2/3), 1) :185
    const bool zwf = nonneg(zQabs);
    const R zcst1 = K<R>::vkarmn * K<R>::grav * zalpha;
    const R ziu2 = M::rcp(zusw2);
    const R zcst2 = zcst1 * R(0.2) * zih * ziu2;
    const R zcst0 = K<R>::rdt * (rNuwl0 + R(1.)) * zih;
    const R zA = zcst0 * zQabs * R(1. / (0.5 * 1025. * 4190.));
    const R zcst3 = -zcst0 * K<R>::vkarmn * zusw * zfLa;
    // zB = zcst3 / PHI(zeta), zeta >= 0 on both branches (:212-217): PHI = 1 + (5z + 4z^2)/(1 + 3z + z^2/4) = (D + N)/D, so
    // zB = zcst3 D / (D + N): one division instead of two
    auto zB_of = [&](R z) -> R {
        const R zD = R(1.) + R(3.) * z + R(0.25) * z * z;
        return M::div(zcst3 * zD, zD + R(5.) * z + R(4.) * z * z);
    };
    R zdTwl_n = zdTwl_b;
    if (zwf) {   // heating: the stability parameter zHwl/L, L from zQabs (:198), does not depend on the iterate
        const R zB = zB_of(zHwl * (zcst1 * zQabs * ziu2 * M::rcp(zRhoCp_w * zusw)));
#pragma unroll 1
        for (int jc = 0; jc < 10; ++jc) {
            zdTwl_n = R(0.5) * (zdTwl_n + zdTwl_b);
            zdTwl_n = vmax(zdTwl_b + zA + zB * zdTwl_n, R(0.));
        }
    } else if (zdTwl_b == R(0.)) {
        // cooling with no warm layer to erode (every cell of a first record, every night after the layer is gone): the ten passes
        // give MAX(0 + zA + zB*0, 0) with zA <= 0, i.e. exactly 0 each time.  Ten square roots and ten divisions not executed.
        zdTwl_n = R(0.);
    } else {
#pragma unroll 1
        for (int jc = 0; jc < 10; ++jc) {
            zdTwl_n = R(0.5) * (zdTwl_n + zdTwl_b);
            const R zB = zB_of(zHwl * M::sqrt(zdTwl_n * zcst2));
            zdTwl_n = vmax(zdTwl_b + zA + zB * zdTwl_n, R(0.));
        }
    }
    dT_wl = zdTwl_n * ztcorr;
}

// ---------------------------------------------------------------- COARE stability functions (mod_common_coare.f90)
// Convective ("free convection") profile function of COARE, mod_common_coare.f90:240-243 (psi_m) and :330-333 (psi_h):
//    c = y**.3333 ,  psi_c = 1.5 LOG((1+c+c*c)/3) - 1.7320508 ATAN((1+2c)/1.7320508) + 1.813799447     (y = |1 - a zeta| >= 1)
// With L = LOG(y) (needed anyway for the .3333 power) and w = 1/c = EXP(-.3333 L) in (0,1]:
//    psi_c = 3 LOG(c) + G(w) = .9999 L + G(w),   G(w) = 1.5 LOG((w*w+w+1)/3) - 1.7320508 (pi/2 - ATAN(1.7320508 w/(w+2))) + 1.813799447
// G is analytic on [0,1]; it is evaluated by its degree-20 near-minimax polynomial in x = 2w-1 (tools/gen_poly.py, fitted
// to the reference's formula WITH its truncated literals; |dG| <= 3.2e-16) instead of a second log and an atan.
AB_TAB double kPsicG[fm::ab_pad4(21)] = {-1.1378018661248832, 1.2857142823699836, -0.15306122324762672, -0.0029154522038313106,
    0.012182424010797529, -0.005319212199747766, 0.0013727273299661688, -0.00012957955060170244,
    -8.228815772429667e-05, 5.619522847612674e-05, -1.9496010425207848e-05, 3.55840718753343e-06,
    4.569112576860549e-07, -6.688538104698402e-07, 2.992044741744515e-07, -7.973350238907201e-08,
    4.974190761282214e-09, 9.732585524385646e-09, -5.605322344298672e-09, 7.426603654517986e-10,
    1.9036812910041772e-10};
// Round 2: the same convective term as a polynomial in ITS OWN log, L = LOG(y): psi_c is analytic in L (nearest singularities at
// L = +-2 pi i), degree 24 on 0 <= L <= LMAX = LOG(1 + 34.15*50) reaches the rounding of its coefficients (3.8e-16 absolute on
// values up to 6.6; tools/gen_poly.py section 7).  One log + 24 FMAs instead of a log, an exponential and the 20 FMAs of G.
// Beyond LMAX (the unclamped zeta of the first guess on a very unstable cell) the form above.
AB_TAB double kPsicL[fm::ab_pad4(25)] = {2.0150925272068325, 2.710495199372137, 0.5590871232010396, -0.16213149529306575,
    0.00260655984246441, 0.021132638463385537, -0.009257773484794939, 0.00030099378954847065,
    0.0016285542947419454, -0.0008130397199026788, 4.4234806548128256e-05, 0.00015150397200069238,
    -8.181957400187382e-05, 5.998120165166437e-06, 1.531936754413038e-05, -8.82516894455308e-06,
    8.203412060926331e-07, 1.6890014539724024e-06, -1.0233606614380625e-06, 4.458163497401233e-08,
    2.2237828297719952e-07, -7.865948843893524e-08, -1.6678750033618155e-08, 1.1728431071594033e-08,
    -7.386244013567441e-10};
// (TAB: the fp32 kernels' piecewise LDS tables; the fp64 kernels read theirs through L1 — ab_gtables.hpp — and reach this function
// only beyond zeta = -50, where the polynomial / closed forms serve)
template <class R, bool TAB = (sizeof(R) == 4 && kPsiTabDefault)> __device__ __forceinline__ R psic_coare(R y)   // y >= 1
{
    using M = Mth<R>;
    const R L = M::log(y);
    if constexpr (sizeof(R) == 8 && TAB) {
        if (L < R(kPsiTabLMax)) return R(psi_tab_eval<kTabPsic>((double)R(L * R(28. / kPsiTabLMax))));
    }
    if constexpr (sizeof(R) == 4 && TAB) {
        if (L < R(kPsiTabLMax)) return R(psi_tab_eval32<2>((float)(L * R(32. / kPsiTabLMax))));
    }
    if constexpr (sizeof(R) == 4) {
        // fp32: G(1/c) by its degree-9 fit (3.4e-8 absolute; tools/gen_poly.py section 5), float literals
        constexpr float g[10] = {-1.1378019f, 1.28571429f, -0.153059546f, -0.00291561207f, 0.0121689119f, -0.00531786575f,
                                 0.00141114178f, -0.000133763491f, -0.000127543533f, 6.20082379e-05f};
        const float x = 2.f * M::exp(-.3333f * L) - 1.f;
        float p = g[9];
#pragma unroll
        for (int i = 8; i >= 0; --i) p = __builtin_fmaf(p, x, g[i]);
        return .9999f * L + p;
    }
    // (fp64 only: in fp32 the hardware exponential costs two slots and a degree-24 polynomial is a loss)
    if (sizeof(R) == 8 && L <= R(7.4433710715553465)) return horner_tab<25, fm::kC_PsicL24>(kPsicL, L * R(2. / 7.4433710715553465) - R(1.));
    const R w = M::exp(R(-.3333) * L);
    return R(.9999) * L + horner_tab<21, fm::kC_PsicG19>(kPsicG, R(2.) * w - R(1.));
}
// The Kansas / Paulson unstable profile functions (mod_common_coare.f90:235-238,326-328 with y = |1 - 15 zeta|;
// mod_blk_ecmwf.f90:462-467,519-523, mod_blk_ncar.f90:350-362, mod_blk_andreas.f90:351-358,402-408 with y = |1 - 16 zeta|):
//    x = y**.25 ,  psi_m = 2 LOG((1+x)/2) + LOG((1+x*x)/2) - 2 ATAN(x) + 0.5 rpi ,  psi_h = 2 LOG((1+x*x)/2)
// as functions of s = LOG(y) >= 0 are analytic with the nearest singularities at s = +-2 pi i: degree-22 polynomials on
// 0 <= s <= SMAX = LOG(801) (zeta >= -50 in every caller's loop; 2.4e-16 / 6e-16 absolute, tools/gen_poly.py section 6).  Where
// both are wanted at the same zeta, one log + 44 FMAs replace two square roots, two logs and an atan with its division.
AB_TAB double kPsikM[fm::ab_pad4(23)] = {1.4034487430788964, 1.962937441601374, 0.5076434534028326, -0.08242303636209851,
    -0.01585625187589632, 0.013840242861725485, -0.0027640918265799284, -0.00102770535119103, 0.0008423134312859619,
    -0.0001578935470667781, -8.491426762767567e-05, 6.264087047136019e-05, -9.793981616650793e-06, -7.69808272687432e-06,
    5.017417775041783e-06, -5.999952584060396e-07, -7.342471370752919e-07, 4.273640809340661e-07, -1.8707465535729833e-08,
    -8.047564473933404e-08, 2.6632819859124706e-08, 6.446912234869585e-09, -3.726709061508255e-09};
AB_TAB double kPsikH[fm::ab_pad4(23)] = {2.301130474750617, 2.813982186689832, 0.3721127642150317, -0.14171504680039929,
    0.01739997587482143, 0.01184423989106276, -0.006994448137656206, 0.0007967564933049481, 0.0008502866902825993,
    -0.00047063994318174133, 3.683870886131965e-05, 7.017844419336646e-05, -3.5162680343753746e-05, 1.2423761509715752e-06,
    6.150938246474367e-06, -2.8078363096058284e-06, -4.417408957567781e-08, 6.0652190760948e-07, -2.2353959952831894e-07,
    -5.548391403673346e-08, 5.3152004586067324e-08, 4.251847861014437e-10, -5.020344151301874e-09};
// psi_m and/or psi_h of the unstable Kansas / Paulson form at y = |1 - a zeta| >= 1.  psi_m (alone or with psi_h): through
// s = LOG(y); psi_h alone: its closed form (a square root and a log) is cheaper than a log and a polynomial.
template <class R, bool TAB = kPsiTabDefault> __device__ __forceinline__ void psik(R y, R *pm, R *ph)
{
    using M = Mth<R>;
    if (pm && sizeof(R) == 8) {   // fp64 only: fp32 square roots, logs and atan are hardware instructions
        const R sl = M::log(y);
        if (sl <= R(6.68586094706836)) {
            if constexpr (sizeof(R) == 8 && TAB) *pm = R(psi_tab_eval<kTabPsikM>((double)R(sl * R(28. / kPsiTabSMax))));   // psi_m: LDS table
            else *pm = horner_tab<23, fm::kC_PsikM21>(kPsikM, sl * R(2. / 6.68586094706836) - R(1.));
            if (ph) *ph = horner_tab<23, fm::kC_PsikH21>(kPsikH, sl * R(2. / 6.68586094706836) - R(1.));
            return;
        }
    }
    if constexpr (sizeof(R) == 4 && TAB) {   // fp32 kernels with the tables: one log, degree-3 pieces
        const R sl = M::log(y);
        if (sl < R(kPsiTabSMax)) {
            const float x32 = (float)(sl * R(32. / kPsiTabSMax));
            if (pm) *pm = R(psi_tab_eval32<0>(x32));
            if (ph) *ph = R(psi_tab_eval32<1>(x32));
            return;
        }
    }
    // psi_h alone, or beyond zeta = -50 (first guess of a very unstable cell, ANDREAS which has no zeta clamp)
    const R x2 = M::sqrt_pos(y);
    if (pm) {
        const R x = M::sqrt_pos(x2);
        const R hx = R(0.5) * (R(1.) + x);
        *pm = M::log(hx * hx * (R(0.5) * (R(1.) + x2))) - R(2.) * M::atan_ge1(x) + R(0.5) * K<R>::rpi;
    }
    if (ph) *ph = R(2.) * M::log(R(0.5) * (R(1.) + x2));
}
// psi_m_coare_sclr :217-254 and psi_h_coare_sclr :305-344 at the same zeta
// SEQ: where psi_m and psi_h are both read from the L1 tables, the second table's loads wait for the first result (sixteen VGPRs of
// coefficients in flight instead of 32: the kernels without the skin schemes run five waves per SIMD on 96 VGPRs)
template <class R, bool SEQ = false> __device__ __forceinline__ void psi_coare(R z, R *pm, R *ph)
{
    AB_REGION("psi_coare");
    using M = Mth<R>;
    if (nonneg(z)) {  // stable: Beljaars & Holtslag (1991)
        const R zc = vmin(R(50.), R(0.35) * z);
        const R t = R(0.6667) * (z - R(14.28)) * M::exp(-zc);
        if (pm) *pm = -(R(1.) + z + t + R(8.525));
        if (ph) {
            const R a = M::abs(R(1.) + R(2.) * z * R(1. / 3.));
            *ph = -(a * M::sqrt_pos(a) + t + R(8.525));
        }
    } else {  // unstable: Kansas / free-convection blend
#ifndef AB_NO_GTABLES
        if constexpr (sizeof(R) == 8) {
            // the WHOLE blended functions of s = LOG(|1 - 15 zeta|), piecewise through L1 (ab_gtables.hpp): one log and two table
            // evaluations on one index instead of three logs, the Kansas table / polynomial, two convective tables and the blend
            const R sl = M::log(M::abs(R(1.) - R(15.) * z));
            if (sl < R(kGPsiSMax)) {
                AB_COUNT("gtab_psi_coare", (pm && ph) ? 19.5 : 12.5);
                GtabPos gp = gtab_pos((double)R(sl * R(kGPsiCoareN / kGPsiSMax)));
                if (pm) {
                    double m = gtab_at(kGPsiCoareM, gp);
#if defined(__HIPCC__) && !defined(AB_FASTMATH_HOST)
                    if (SEQ && ph) asm volatile("" : "+v"(gp.i), "+v"(m));
#endif
                    *pm = R(m);
                }
                if (ph) *ph = R(gtab_at(kGPsiCoareH, gp));
                return;
            }
        }
#endif
        R zf = z * z;
        zf = M::div(zf, R(1.) + zf);
        R psik_m = R(0.), psik_h = R(0.);
        psik<R, (sizeof(R) == 4 && kPsiTabDefault)>(M::abs(R(1.) - R(15.) * z), pm ? &psik_m : nullptr, ph ? &psik_h : nullptr);
        if (pm) *pm = (R(1.) - zf) * psik_m + zf * psic_coare(M::abs(R(1.) - R(10.15) * z));
        if (ph) *ph = (R(1.) - zf) * psik_h + zf * psic_coare(M::abs(R(1.) - R(34.15) * z));
    }
}
template <class R> __device__ __forceinline__ R psi_h_coare(R z) { R h; psi_coare<R>(z, nullptr, &h); return h; }
template <class R> __device__ __forceinline__ R psi_m_coare(R z) { R m; psi_coare<R>(z, &m, nullptr); return m; }

// charn_coare3p6_sclr mod_blk_coare3p6.f90:417-432 (Edson 2013 Eq.13)
template <class R> __device__ __forceinline__ R charn_coare3p6(R w)
{
    return vmax(vmin(R(0.0017) * w - R(0.005), R(0.028)), R(0.));
}
// charn_coare3p0 mod_blk_coare3p0.f90:420-447 (Hare 1999 ramp 10..18 m/s)
template <class R> __device__ __forceinline__ R charn_coare3p0(R w)
{
    if (!nonneg(w - R(10.))) return R(0.011);
    if (nonneg(w - R(18.))) return R(0.018);
    return R(0.011) + R(0.018 - 0.011) * (w - R(10.)) * R(1. / (18. - 10.));
}

// FIRST_GUESS_COARE_SCLR mod_common_coare.f90:33-179 (shared by COARE 3.0/3.6 and ECMWF)
template <class R> struct Heights {  // wave-uniform, prepared on the host
    R zt, zu, log_zt, log_zu, log_10, log_ztu, log_zu10, fg_ca, inv_zu, zt_o_zu;
    int zt_eq_zu;  // ABS(zu-zt) < 0.01
};
template <class R, class A = R, bool SEQ = false>
__device__ __forceinline__ void first_guess_coare(const Heights<R> &h, A psst, A t_zt, A pssq, A q_zt, R U_zu,
                                                  R pcharn, R &pus, R &pts, R &pqs, A &t_zu, A &q_zu, R &Ubzu,
                                                  R &pz0)
{
    AB_REGION("first_guess_coare");
    using M = Mth<R>;
    const R vk = K<R>::vkarmn;
    t_zu = vmax(t_zt, A(180.));
    q_zu = vmax(q_zt, A(1.e-6));
    const R zc_b = R(0.004 * 600. * 1.2 * 1.2 * 1.2);
    R zdt = sfloor(R(t_zu - psst), R(1.E-09));
    R zdq = sfloor(R(q_zu - pssq), R(1.E-12));
    const R zNu_a = visc_air(R(t_zu));
    const R zUb = M::sqrt_pos(U_zu * U_zu + R(0.25));
    R zus = h.fg_ca * zUb;  // zc_a = 0.035*LOG(10/z0)/LOG(zu/z0), z0 = 1e-4 :107
    R zz0 = pcharn * zus * zus * R(1. / 9.8) + M::div(R(0.11) * zNu_a, zus);
    zz0 = vmin(vmax(M::abs(zz0), R(1.E-8)), R(1.));
    const R zlog_z0 = M::log(zz0);
    const R zdl = h.log_zu - zlog_z0;
    // z0t = 10/exp(kappa/(0.00115 (ln10-ln z0)/kappa)) :130, clamped to [1e-8,1]; only its log is used
    const R zlog_z0t = vmin(vmax(h.log_10 - M::div(vk, R(0.00115) * ((h.log_10 - zlog_z0) * K<R>::inv_vk)), R(-18.420680743952367)), R(0.));
    const R zRib = ri_bulk<R, A>(h.zu, psst, t_zu, pssq, q_zu, zUb);
    // kappa^2/(Cd (ln zt - ln z0t)) Ri, Cd = (kappa/(ln zu - ln z0))^2  :126,138-139
    const R zcc_ri = M::div(zdl * zdl, h.log_zt - zlog_z0t) * zRib;
    R zzeta_u;
    if (nonneg(zRib)) zzeta_u = zcc_ri + R(27. / 9.) * zRib * zRib;
    else zzeta_u = M::div(zcc_ri, R(1.) + zRib * (-zc_b * h.inv_zu));
    R psm, psh;
    psi_coare<R, SEQ>(zzeta_u, &psm, &psh);
    zus = vmax(M::div(zUb * vk, zdl - psm), R(1.E-9));
    const R ztmp = M::div(vk, h.log_zu - zlog_z0t - psh);
    R zts = zdt * ztmp;
    R zqs = zdq * ztmp;
    if (!h.zt_eq_zu) {
        const R zzeta_t = zzeta_u * h.zt_o_zu;
        const R zprf = h.log_ztu + psh - psi_h_coare<R>(zzeta_t);
        t_zu = t_zt - A(zts * K<R>::inv_vk * zprf);
        q_zu = q_zt - A(zqs * K<R>::inv_vk * zprf);
        q_zu = nonneg(q_zu) ? q_zu : A(0.) * q_zu;
        zdt = sfloor(R(t_zu - psst), R(1.E-09));
        zdq = sfloor(R(q_zu - pssq), R(1.E-12));
        zts = zdt * ztmp;
        zqs = zdq * ztmp;
    }
    pus = zus; pts = zts; pqs = zqs; Ubzu = zUb;
    zz0 = pcharn * zus * zus * R(1. / 9.8) + M::div(R(0.11) * zNu_a, zus);
    pz0 = vmin(vmax(M::abs(zz0), R(1.E-8)), R(1.));
}

// ---------------------------------------------------------------- per-cell inputs / outputs of a TURB_* routine
// ANCHORS.  Temperatures and humidities — SST, theta, T_s, q, q_s — are the quantities whose DIFFERENCES drive the fluxes
// (theta_zu - T_s ~ 1 K of 300, q_zu - q_s ~ 1e-3 of 1e-2): they are carried in the anchor type A, everything else (profile
// functions, roughness lengths, scales, skin increments) in R.  A = R for the fp64 and the fp32 kernels.  The mixed mode
// (AB_F32_MIXED: R = float, A = double) keeps the anchors, their differences and q_sat in fp64 and lets the hardware fp32
// transcendentals do the rest: the fp32 kernels' speed without their 1e-3 errors (mod_blk_ecmwf.f90:556-561 warns about exactly
// these differences in single precision).
template <class R, class A = R> struct CellIn {
    A sst, theta_zt, ssq, q_zt;            // after the pre-processing of aerobulk_compute :99-126
    R wnd, slp;
    R qsw, rlw;                            // (1-albedo)*rad_sw, rad_lw   (skin only)
};
template <class R, class A = R> struct CellOut {
    R Cd, Ch, Ce;
    A t_zu, q_zu;
    R Ubzu;
    A T_s, q_s;
    // OPTIONAL outputs of the TURB_* routines (CdN ChN CeN xz0 xu_star xL xUN10 pdT_cs pdT_wl pHz_wl), filled only by the
    // DIAG instantiations (ab_session_set_diagnostics); dead code otherwise
    R CdN, ChN, CeN, z0, us, L, UN10, dT_cs, dT_wl, Hz_wl;
};

// ---------------------------------------------------------------- TURB_COARE3P6 / turb_coare3p0
// mod_blk_coare3p6.f90:123-413, mod_blk_coare3p0.f90:54-358.  V36 selects the version; SKIN is a bit mask: bit 0 =
// l_use_cs (cool skin), bit 1 = l_use_wl (warm layer).  aerobulk_compute always switches both on together
// (mod_aerobulk_compute.f90:133,144: SKIN = 3); callers of TURB_* may choose either (ab_session_turb).
constexpr int kSkinCS = 1, kSkinWL = 2, kSkinBoth = 3;
// park / pstride: optional per-lane scratch words park[i * pstride], i = 0,1,2,5,6 (LDS slots of the caller's tile).  The warm-layer state beyond dT_wl and the
// two WL_COARE constants are only touched by the live WL_COARE calls (jit = 1 and the divisors of nb_iter): parked there they
// do not hold ten VGPRs across the rest of the iteration (the COARE + skin kernels sit at the 128-VGPR limit).
// CSGLDS: the caller's kernel keeps the cool skin's g(u) table in LDS (flux_kernel: csg_table_fill)
template <class R, bool V36, int SKIN, bool DIAG = false, class A = R, bool CSGLDS = false>
__device__ __forceinline__ void turb_coare(const Heights<R> &h, const CellIn<R, A> &in, int nb_iter, R (&wl)[4],
                                           bool dawn, CellOut<R, A> &o, lds_vptr<R> park = nullptr, int pstride = 0)
{
    using M = Mth<R>;
    const R vk = K<R>::vkarmn;
    const R Beta0 = V36 ? R(1.2) : R(1.25);
    const R zi0 = R(600.), zeta_max = R(50.);
    const R zUzu = in.wnd;
    const A xSST = in.sst;
    A T_s = in.sst, q_s = in.ssq;
    R zalpha = R(0.);
    WlCoareCell<R> wc{R(0.), R(0.), dawn};
    constexpr bool CS = (SKIN & kSkinCS) != 0, WL = (SKIN & kSkinWL) != 0;
    if (SKIN) {
        if (CS) T_s = T_s - A(0.25);                                   // :274
        q_s = rounded(K<A>::rdct_qsat_salt * q_sat(vmax(T_s, A(200.)), A(in.slp)));  // :275
        zalpha = alpha_sw(R(xSST));                                    // hoisted from CS_COARE :81 / WL_COARE :153
        if (WL) {
            const R Rich0 = R(0.65);
            wc.zcd1 = M::sqrt(M::div(R(2.) * Rich0 * K<R>::rCp0_w, zalpha * K<R>::grav * K<R>::rho0_w));   // :155
            wc.zcd2 = M::sqrt(R(2.) * zalpha * K<R>::grav * R(1. / (0.65 * 1025.)))
                      * R(1. / 271219.5770957547);                    // / rCp0_w**1.5 :156
        }
    }
    const bool parked = WL && park != nullptr;
    if (parked) {
        park[0] = wl[1]; park[pstride] = wl[2]; park[2 * pstride] = wl[3];
        park[5 * pstride] = wc.zcd1; park[6 * pstride] = wc.zcd2;
    }
    R zus, zts, zqs, Ubzu, zz0;
    A t_zu, q_zu;
    first_guess_coare<R, A, SKIN == 0>(h, T_s, in.theta_zt, q_s, in.q_zt, zUzu, V36 ? charn_coare3p6(zUzu) : charn_coare3p0(zUzu),
                                       zus, zts, zqs, t_zu, q_zu, Ubzu, zz0);
    R zlog_z0 = M::log(zz0);
    const R znu_a = visc_air(R(V36 ? t_zu : in.theta_zt));  // 3p6 :294 (first-guess t_zu) vs 3p0 :237 (t_zt)
    const R zlog_nu = M::log(znu_a);
    R zdt = sfloor(R(t_zu - T_s), R(1.E-09));
    R zdq = sfloor(R(q_zu - q_s), R(1.E-12));
    R zdT_cs = R(0.);
    R d_1oL = R(0.), d_lz0t = R(0.);   // DIAG only

#pragma unroll 1
    for (int jit = 1; jit <= nb_iter; ++jit) {
        const R zus2 = zus * zus;
        const R z1oL = one_on_l(R(t_zu), R(q_zu), zus, zts, zqs);      // :307-308 (second clamp is idempotent)
        // gustiness :311-313: Ug^2 = Beta0^2 u*^2 (max(-zi0/(kappa L),0))^(2/3)
        const R zg = vmax(-zi0 * z1oL * K<R>::inv_vk, R(0.));
        const R zcb = M::cbrt(zg);
        const R zgust2 = Beta0 * Beta0 * zus2 * (zcb * zcb);
        Ubzu = vmax(M::sqrt(zUzu * zUzu + zgust2), R(0.2));
        const R zzta_u = sclamp(h.zu * z1oL, zeta_max);                // :317-318
        // roughness lengths :328-336 (3p0 :270-278)
        const R zUn10 = zus * K<R>::inv_vk * (h.log_10 - zlog_z0);
        const R zlog_us = M::log(zus);
        zz0 = (V36 ? charn_coare3p6(zUn10) : charn_coare3p0(zUn10)) * zus2 * R(1. / 9.8) + M::div(R(0.11) * znu_a, zus);
        zz0 = vmin(vmax(M::abs(zz0), R(1.E-9)), R(1.));
        zlog_z0 = M::log(zz0);
        // z0t = min(1.6e-4, 5.8e-5 Rr^-0.72) (3p6 :333-334) | min(1.1e-4, 5.5e-5 Rr^-0.6) (3p0 :275-276), Rr = z0 u*/nu,
        // clamped to [1e-9,1] (:335): only LOG(z0t) is used (:336,340), so it is formed in the log domain
        const R zlog_rr = zlog_nu - zlog_z0 - zlog_us;                 // ln(nu/(z0 u*))
        const R zlog_z0t = vmax(V36 ? vmin(R(-8.740336742730447), R(-9.755067547417855) + R(0.72) * zlog_rr)
                                    : vmin(R(-9.115030192171858), R(-9.808177372731803) + R(0.6) * zlog_rr),
                                R(-20.72326583694641));
        if (DIAG) { d_1oL = z1oL; d_lz0t = zlog_z0t; }
        // turbulent scales :339-344
        R psm, psh;
        psi_coare<R, SKIN == 0>(zzta_u, &psm, &psh);
        R ztmp1 = M::div(vk, h.log_zu - zlog_z0t - psh);
        zts = zdt * ztmp1;
        zqs = zdq * ztmp1;
        zus = vmax(M::div(Ubzu * vk, h.log_zu - zlog_z0 - psm), R(1.E-9));
        if (!h.zt_eq_zu) {                                             // :346-351 (3p0 :289-291 with zm_ztzu = 1)
            const R zzta_t = sclamp(h.zt * z1oL, zeta_max);
            ztmp1 = h.log_zt - h.log_zu + psh - psi_h_coare<R>(zzta_t);
            t_zu = in.theta_zt - A(zts * K<R>::inv_vk * ztmp1);
            q_zu = in.q_zt - A(zqs * K<R>::inv_vk * ztmp1);
        } else if (!V36) {                                             // 3p0: t_zu = t_zt - 0*...  (drops the 180 K floor)
            t_zu = in.theta_zt;
            q_zu = in.q_zt;
        }
        if (CS) {
            R zQns, zTau, zQlat;
            update_qnsol_tau<R, A>(h.zu, T_s, q_s, t_zu, q_zu, zus, zts, zqs, zUzu, Ubzu, in.slp, in.rlw, zQns, zTau,
                                   zQlat);                             // :355-356
            zdT_cs = cool_skin<R, true, CSGLDS>(in.qsw, zQns, zus, zalpha, zQlat);  // :358
            T_s = xSST + A(zdT_cs);
            if (WL) T_s = T_s + A(wl[0]);                              // :360-361
            // with the warm layer on, this q_s is only read by the UPDATE_QNSOL_TAU of a live WL_COARE call (below)
            if (!WL || (nb_iter % jit) == 0) q_s = rounded(K<A>::rdct_qsat_salt * q_sat(vmax(T_s, A(200.)), A(in.slp)));
        }
        if (WL) {
            // WL_COARE is called with iwait = MOD(nb_iter,jit) (:370) and writes its state (dT_wl, Hz_wl, Qnt_ac, Tau_ac)
            // only when iwait == 0 (mod_skin_coare.f90:239-248); everything else in it is local.  For the other
            // iterations (jit = 2,3,4 of 5) the call and the UPDATE_QNSOL_TAU feeding it have no effect: skipped.
            if ((nb_iter % jit) == 0) {
                R zQns, zTau, zQlat;
                update_qnsol_tau<R, A>(h.zu, T_s, q_s, t_zu, q_zu, zus, zts, zqs, zUzu, Ubzu, in.slp, in.rlw, zQns, zTau,
                                       zQlat);                         // :367-368
                if (parked) {
                    wl[1] = park[0]; wl[2] = park[pstride]; wl[3] = park[2 * pstride];
                    wc.zcd1 = park[5 * pstride]; wc.zcd2 = park[6 * pstride];
                }
                wl_coare(wl, wc, in.qsw, zQns, zTau, true);            // :370
                if (parked) { park[0] = wl[1]; park[pstride] = wl[2]; park[2 * pstride] = wl[3]; }
            }
            T_s = xSST + A(wl[0]);
            if (CS) T_s = T_s + A(zdT_cs);                             // :373-374
            q_s = rounded(K<A>::rdct_qsat_salt * q_sat(vmax(T_s, A(200.)), A(in.slp)));
        }
        if (!V36 || SKIN || !h.zt_eq_zu) {                             // :378-381 (3p0 :317-318 unconditional)
            zdt = sfloor(R(t_zu - T_s), R(1.E-09));
            zdq = sfloor(R(q_zu - q_s), R(1.E-12));
        }
    }
    if (parked) { wl[1] = park[0]; wl[2] = park[pstride]; wl[3] = park[2 * pstride]; }
    const R ztmp0 = M::div(zus, Ubzu);                                 // :386-389
    o.Cd = vmax(ztmp0 * ztmp0, K<R>::Cx_min);
    o.Ch = vmax(M::div(ztmp0 * zts, zdt), K<R>::Cx_min);
    o.Ce = vmax(M::div(ztmp0 * zqs, zdq), K<R>::Cx_min);
    o.t_zu = t_zu; o.q_zu = q_zu; o.Ubzu = Ubzu; o.T_s = T_s; o.q_s = q_s;
    if (DIAG) {   // optional outputs :392-407 (3p0 :337-352)
        const R zi = M::rcp(h.log_zu - zlog_z0);
        o.CdN = vmax(K<R>::vkarmn2 * zi * zi, K<R>::Cx_min);
        o.ChN = vmax(M::div(K<R>::vkarmn2 * zi, h.log_zu - d_lz0t), K<R>::Cx_min);
        o.CeN = o.ChN;
        o.z0 = zz0; o.us = zus; o.L = M::rcp(d_1oL); o.UN10 = zus * K<R>::inv_vk * (h.log_10 - zlog_z0);
        o.dT_cs = zdT_cs; o.dT_wl = (SKIN & kSkinWL) ? wl[0] : R(0.); o.Hz_wl = (SKIN & kSkinWL) ? wl[1] : R(0.);
    }
}

// ---------------------------------------------------------------- ECMWF (mod_blk_ecmwf.f90)
// psi_m_ecmwf_scl :441-477, psi_h_ecmwf_scl :498-533, cap_zeta :551-564
template <class R> __device__ __forceinline__ void psi_ecmwf(R pz, R *pm, R *ph)
{
    AB_REGION("psi_ecmwf");
    using M = Mth<R>;
    const R zc = R(5. / 0.35);
    const R z = vmin(vmax(pz, R(-50.)), R(5.));
    if (nonneg(z)) {
        const R t = R(-2. / 3.) * (z - zc) * M::exp(R(-0.35) * z);
        if (pm) *pm = t - z - R(2. / 3.) * zc;
        if (ph) {
            const R a = M::abs(R(1.) + R(2. / 3.) * z);
            *ph = t - a * M::sqrt_pos(a) - R(2. / 3.) * zc + R(1.);
        }
    } else {
        psik<R>(M::abs(R(1.) - R(16.) * z), pm, ph);
    }
}
template <class R> __device__ __forceinline__ R psi_m_ecmwf(R z) { R m; psi_ecmwf<R>(z, &m, nullptr); return m; }
template <class R> __device__ __forceinline__ R psi_h_ecmwf(R z) { R v; psi_ecmwf<R>(z, nullptr, &v); return v; }
// The same functions at the roughness heights: turb_ecmwf evaluates psi_m(z0/L), psi_h(z0t/L), psi_h(z0q/L) in every iteration
// (:276,290-292), arguments of 1e-6 .. 1e-4.  There the closed forms are O(1) terms cancelling to O(zeta) — two square roots,
// a log, an atan with its division — for a number a degree-5/6 polynomial gives to 1e-18 absolute (each function is analytic
// on its side of 0; fitted on |zeta| <= 1e-3 in t = 1000 zeta, tools/gen_psi_small.py).  Beyond 1e-3 the closed forms.
AB_TAB double kPsiMU[fm::ab_pad4(7)] = {-6.094914512234566e-17, -0.003999999999999972, -1.9999999999555597e-05, -1.599999973300714e-07,
                                        -1.5599923538873374e-09, -1.69615384228405e-11, -1.8975349598658237e-13};
AB_TAB double kPsiHU[fm::ab_pad4(7)] = {9.367832564752625e-19, -0.007999999999999908, -4.799999999852992e-05, -4.266666578341352e-07,
                                        -4.4799747034570265e-09, -5.1572336558726693e-11, -6.034358874849738e-13};
AB_TAB double kPsiMS[fm::ab_pad4(6)] = {-2.6116088597091867e-26, -0.005, 8.166666666666666e-07, -1.0888888888879528e-10,
                                        1.071874981948049e-14, -8.335200884605216e-19};
AB_TAB double kPsiHS[fm::ab_pad4(6)] = {2.665432043173813e-25, -0.005, 6.500000000000002e-07, -9.037037037132576e-11,
                                        6.089122213161426e-15, 7.078908223179431e-19};
template <class R> __device__ __forceinline__ R psi_m_ecmwf_z0(R z)
{
    AB_REGION("psi_ecmwf_z0");
    if (Mth<R>::abs(z) > R(1.e-3)) return psi_m_ecmwf<R>(z);
    const R t = z * R(1000.);
    if (sizeof(R) == 4)   // fp32: the first three terms (the fourth is below 2e-9), literal coefficients
        return nonneg(z) ? t * (R(-0.005) + t * (R(8.166666667e-07) + t * R(-1.0888889e-10)))
                         : t * (R(-0.004) + t * (R(-2.0e-05) + t * R(-1.6e-07)));
    return nonneg(z) ? horner_tab<6>(kPsiMS, t) : horner_tab<7>(kPsiMU, t);
}
template <class R> __device__ __forceinline__ R psi_h_ecmwf_z0(R z)
{
    AB_REGION("psi_ecmwf_z0");
    if (Mth<R>::abs(z) > R(1.e-3)) return psi_h_ecmwf<R>(z);
    const R t = z * R(1000.);
    if (sizeof(R) == 4)
        return nonneg(z) ? t * (R(-0.005) + t * (R(6.5e-07) + t * R(-9.037037e-11)))
                         : t * (R(-0.008) + t * (R(-4.8e-05) + t * R(-4.2666667e-07)));
    return nonneg(z) ? horner_tab<6>(kPsiHS, t) : horner_tab<7>(kPsiHU, t);
}

// turb_ecmwf :63-383
template <class R, int SKIN, bool DIAG = false, class A = R>
__device__ __forceinline__ void turb_ecmwf(const Heights<R> &h, const CellIn<R, A> &in, int nb_iter, R (&wl)[4],
                                           CellOut<R, A> &o)
{
    using M = Mth<R>;
    const R vk = K<R>::vkarmn;
    const R charn0 = R(0.018), zi0 = R(1000.), alpha_M = R(0.11), alpha_H = R(0.40), alpha_Q = R(0.62);
    const R zm_ztzu = h.zt_eq_zu ? R(0.) : R(1.);
    const R zUzu = in.wnd;
    const A zSST = in.sst;
    A zT_s = in.sst, zq_s = in.ssq;
    R zalpha = R(0.);
    constexpr bool CS = (SKIN & kSkinCS) != 0, WL = (SKIN & kSkinWL) != 0;
    if (SKIN) {
        if (CS) zT_s = zT_s - A(0.25);                                  // :214
        zq_s = rounded(K<A>::rdct_qsat_salt * q_sat(vmax(zT_s, A(200.)), A(in.slp)));
        zalpha = alpha_sw(R(zSST));
    }
    R zus, zts, zqs, zUbzu, zz0;
    A zt_zu, zq_zu;
    first_guess_coare<R, A>(h, zT_s, in.theta_zt, zq_s, in.q_zt, zUzu, charn0, zus, zts, zqs, zt_zu, zq_zu, zUbzu, zz0);
    R zlog_z0 = M::log(zz0);
    const R znu_a = visc_air(R(in.theta_zt));                           // :238
    R zdt = sfloor(R(zt_zu - zT_s), R(1.E-09));
    R zdq = sfloor(R(zq_zu - zq_s), R(1.E-12));
    R z1oL = one_on_l(R(zt_zu), R(zq_zu), zus, zts, zqs);               // :245
    R zzeta_u = h.zu * z1oL;
    // :249  z0t = 1/(0.1 exp(kappa/(0.00115/(kappa/(ln10 - ln z0)))))
    // :249  z0t = 1/(0.1 exp(kappa/(0.00115/(kappa/(ln10 - ln z0))))) clamped to [1e-9,1]; log and value both needed
    R zlog_z0t = vmin(vmax(h.log_10 - M::div(vk, M::div(R(0.00115), M::div(vk, h.log_10 - zlog_z0))), R(-20.72326583694641)), R(0.));
    R zz0t = M::exp(zlog_z0t);
    R zpsi_m_u, zpsi_h_u;
    psi_ecmwf<R>(zzeta_u, &zpsi_m_u, &zpsi_h_u);
    R zFm = h.log_zu - zlog_z0 - zpsi_m_u + psi_m_ecmwf_z0<R>(zz0 * z1oL);   // :253
    R zFh = h.log_zu - zlog_z0t - zpsi_h_u + psi_h_ecmwf_z0<R>(zz0t * z1oL); // :255
    R zlog_z0q = R(0.), zpsi_h_z0q = R(0.), zdT_cs = R(0.);
    WlEcmwfCell<R> wlc{R(0.), R(0.)};
    if (WL) wlc = wl_ecmwf_cell(wl[1]);

#pragma unroll 1
    for (int jit = 1; jit <= nb_iter; ++jit) {
        const R zRib = ri_bulk<R, A>(h.zu, zT_s, zt_zu, zq_s, zq_zu, zUbzu);  // :261 (previous Ub, T_s, q_s)
        z1oL = sclamp(M::div(zRib * zFm * zFm, zFh) * h.inv_zu, R(200.));  // :264-266
        zzeta_u = h.zu * z1oL;
        psi_ecmwf<R>(zzeta_u, &zpsi_m_u, &zpsi_h_u);                    // :269-270
        const R zpsi_h_t = psi_h_ecmwf<R>(h.zt * z1oL);                 // :272-273
        zFm = h.log_zu - zlog_z0 - zpsi_m_u + psi_m_ecmwf_z0<R>(zz0 * z1oL);  // :276
        zus = M::div(zUbzu * vk, zFm);                                  // :279
        const R zus2 = zus * zus;
        R ztmp0 = M::div(znu_a, zus);
        zz0 = vmin(M::abs(alpha_M * ztmp0 + charn0 * zus2 * R(1. / 9.8)), R(0.001));  // :282-284
        zz0t = vmin(M::abs(alpha_H * ztmp0), R(0.001));
        const R zz0q = vmin(M::abs(alpha_Q * ztmp0), R(0.001));
        zlog_z0 = M::log(zz0);
        const R zlog_nuus = M::log(M::abs(ztmp0));                      // z0t, z0q share ln(nu/u*) :287-288
        zlog_z0t = vmin(R(-0.916290731874155) + zlog_nuus, R(-6.907755278982137));
        zlog_z0q = vmin(R(-0.4780358009429998) + zlog_nuus, R(-6.907755278982137));
        const R zpsi_m_z0 = psi_m_ecmwf_z0<R>(zz0 * z1oL);              // :290-292
        const R zpsi_h_z0t = psi_h_ecmwf_z0<R>(zz0t * z1oL);
        zpsi_h_z0q = psi_h_ecmwf_z0<R>(zz0q * z1oL);
        // gustiness :296-298 (Beta0 = 1)
        const R zcb = M::cbrt(vmax(-zi0 * z1oL * K<R>::inv_vk, R(0.)));
        zUbzu = vmax(M::sqrt(zUzu * zUzu + zus2 * (zcb * zcb)), R(0.2));
        // t*, q* and height adjustment :303-313
        ztmp0 = zpsi_h_u - zpsi_h_z0t;
        R ztmp1 = M::div(vk, h.log_zu - zlog_z0t - ztmp0);
        zts = zdt * ztmp1;
        ztmp1 = h.log_ztu + ztmp0 - zpsi_h_t + zpsi_h_z0t;
        zt_zu = in.theta_zt - A(zm_ztzu * zts * K<R>::inv_vk * ztmp1);
        ztmp0 = zpsi_h_u - zpsi_h_z0q;
        ztmp1 = M::div(vk, h.log_zu - zlog_z0q - ztmp0);
        zqs = zdq * ztmp1;
        ztmp1 = h.log_ztu + ztmp0 - zpsi_h_t + zpsi_h_z0q;
        zq_zu = vmax(in.q_zt - A(zm_ztzu * zqs * K<R>::inv_vk * ztmp1), A(0.));
        zFm = h.log_zu - zlog_z0 - zpsi_m_u + zpsi_m_z0;                // :316-317
        zFh = h.log_zu - zlog_z0t - zpsi_h_u + zpsi_h_z0t;
        if (CS) {
            R zQns, zTau, zQlat;
            update_qnsol_tau<R, A>(h.zu, zT_s, zq_s, zt_zu, zq_zu, zus, zts, zqs, zUzu, zUbzu, in.slp, in.rlw, zQns, zTau,
                                   zQlat);                              // :321-322
            zdT_cs = cool_skin<R, false>(in.qsw, zQns, zus, zalpha, R(0.));  // :324
            zT_s = zSST + A(zdT_cs);
            if (WL) zT_s = zT_s + A(wl[0]);
            zq_s = rounded(K<A>::rdct_qsat_salt * q_sat(vmax(zT_s, A(200.)), A(in.slp)));
        }
        if (WL) {
            R zQns, zTau, zQlat;
            update_qnsol_tau<R, A>(h.zu, zT_s, zq_s, zt_zu, zq_zu, zus, zts, zqs, zUzu, zUbzu, in.slp, in.rlw, zQns, zTau,
                                   zQlat);                              // :333-334
            wl_ecmwf(wl[0], wl[1], wlc, in.qsw, zQns, zus, zalpha);     // :335
            zT_s = zSST + A(wl[0]);
            if (CS) zT_s = zT_s + A(zdT_cs);
            zq_s = rounded(K<A>::rdct_qsat_salt * q_sat(vmax(zT_s, A(200.)), A(in.slp)));
        }
        zdt = sfloor(R(zt_zu - zT_s), R(1.E-09));                       // :342-343
        zdq = sfloor(R(zq_zu - zq_s), R(1.E-12));
    }
    const R zFq = h.log_zu - zlog_z0q - zpsi_h_u + zpsi_h_z0q;          // :356-359
    const R ziFm = M::div(K<R>::vkarmn2, zFm);
    o.Cd = vmax(M::div(ziFm, zFm), K<R>::Cx_min);
    o.Ch = vmax(M::div(ziFm, zFh), K<R>::Cx_min);
    o.Ce = vmax(M::div(ziFm, zFq), K<R>::Cx_min);
    o.t_zu = zt_zu; o.q_zu = zq_zu; o.Ubzu = zUbzu; o.T_s = zT_s; o.q_s = zq_s;
    if (DIAG) {   // optional outputs :362-377
        const R zi = M::rcp(h.log_zu - zlog_z0);
        o.CdN = vmax(K<R>::vkarmn2 * zi * zi, K<R>::Cx_min);
        o.ChN = vmax(M::div(K<R>::vkarmn2 * zi, h.log_zu - zlog_z0t), K<R>::Cx_min);
        o.CeN = o.ChN;
        o.z0 = zz0; o.us = zus; o.L = M::rcp(z1oL); o.UN10 = zus * K<R>::inv_vk * (h.log_10 - zlog_z0);
        o.dT_cs = zdT_cs; o.dT_wl = (SKIN & kSkinWL) ? wl[0] : R(0.); o.Hz_wl = (SKIN & kSkinWL) ? wl[1] : R(0.);
    }
}

// ---------------------------------------------------------------- NCAR (mod_blk_ncar.f90, Large & Yeager 2004/2008)
// cd_n10_ncar_sclr :244-271
template <class R> __device__ __forceinline__ R cd_n10_ncar(R zw)
{
    R r;
    if (nonneg(zw - R(33.))) {
        r = R(1.e-3) * R(2.34);
    } else {
        R zw6 = zw * zw * zw;
        zw6 = zw6 * zw6;
        r = R(1.e-3) * (Mth<R>::div(R(2.7), zw) + R(0.142) + zw * R(1. / 13.09) - R(3.14807E-10) * zw6);
    }
    return vmax(r, K<R>::Cx_min);
}
// psi_m_ncar_sclr :333-363 / psi_h_ncar_sclr :379-407
template <class R> __device__ __forceinline__ void psi_ncar(R z, R *pm, R *ph)
{
    AB_REGION("psi_ncar");
    using M = Mth<R>;
    if (nonneg(z)) {
        if (pm) *pm = R(-5.) * z;
        if (ph) *ph = R(-5.) * z;
    } else {   // x2 = MAX(SQRT(ABS(1 - 16 zeta)), 1) (:350,395): the MAX never binds for zeta < 0
        psik<R, false>(M::abs(R(1.) - R(16.) * z), pm, ph);   // (the direct NCAR kernel fills no psi table)
    }
}
// turb_ncar :57-240
template <class R, bool DIAG = false, class A = R>
__device__ __forceinline__ void turb_ncar(const Heights<R> &h, const CellIn<R, A> &in, int nb_iter, CellOut<R, A> &o)
{
    using M = Mth<R>;
    const R vk = K<R>::vkarmn;
    const A sst = in.sst, ssq = in.ssq;
    const R Ubzu = vmax(R(0.5), in.wnd);                                // :148
    bool stab = nonneg(virt_temp(in.theta_zt, in.q_zt) - virt_temp(sst, ssq));  // :158
    R zCdN = cd_n10_ncar(Ubzu);
    R zsqrt_CdN = M::sqrt_pos(zCdN);
    R Cd = zCdN;
    R Ce = vmax(R(1.e-3) * (R(34.6) * zsqrt_CdN), K<R>::Cx_min);         // ce_n10 :321
    R Ch = vmax(R(1.e-3) * zsqrt_CdN * (stab ? R(18.) : R(32.7)), K<R>::Cx_min);  // ch_n10 :301
    R zsqrt_Cd = zsqrt_CdN;
    A t_zu = vmax(in.theta_zt, A(180.));
    A q_zu = vmax(in.q_zt, A(1.e-6));
    R d_us = R(0.), d_1oL = R(0.), d_un10 = R(0.), d_chn = R(0.), d_cen = R(0.);   // DIAG only
#pragma unroll 1
    for (int jit = 1; jit <= nb_iter; ++jit) {
        const R zdt = R(t_zu - sst);                                    // :177-178 (not floored)
        const R zdq = R(q_zu - ssq);
        const R zus = zsqrt_Cd * Ubzu;
        const R zisq = M::rcp(zsqrt_Cd);
        const R zts = Ch * zisq * zdt;
        const R zqs = Ce * zisq * zdq;
        const R z1oL = one_on_l(R(t_zu), R(q_zu), zus, zts, zqs);
        const R zeta_u = sclamp(h.zu * z1oL, R(10.));                   // :189-190
        R psm, psh;
        psi_ncar<R>(zeta_u, &psm, &psh);
        if (!h.zt_eq_zu) {                                              // :193-200
            const R zeta_t = sclamp(h.zt * z1oL, R(10.));
            R psht;
            psi_ncar<R>(zeta_t, nullptr, &psht);
            const R ztmp = h.log_ztu + psh - psht;
            t_zu = in.theta_zt - A(zts * K<R>::inv_vk * ztmp);
            q_zu = in.q_zt - A(zqs * K<R>::inv_vk * ztmp);
            q_zu = vmax(A(0.), q_zu);
        }
        // UN10_from_CD mod_phymbl.f90:1545 with z0_from_Cd :1346:
        //   sqrt(Cd) Ub/kappa * ln(10/(zu exp(-(kappa/sqrt(Cd)+psi)))) = sqrt(Cd) Ub/kappa * (ln(10/zu) + kappa/sqrt(Cd) + psi)
        // (zsqrt_Cd == SQRT(Cd) at this point: first guess :168, then :216)
        const R zUn10 = vmax(R(0.25), zsqrt_Cd * Ubzu * K<R>::inv_vk * (-h.log_zu10 + (vk * zisq + psm)));  // :207
        zCdN = cd_n10_ncar(zUn10);
        zsqrt_CdN = M::sqrt_pos(zCdN);
        R ztmp = R(1.) + zsqrt_CdN * K<R>::inv_vk * (h.log_zu10 - psm);  // :213
        Cd = vmax(M::div(zCdN, ztmp * ztmp), K<R>::Cx_min);
        zsqrt_Cd = M::sqrt_pos(Cd);
        const R ziN = M::rcp(zsqrt_CdN);
        ztmp = (h.log_zu10 - psh) * K<R>::inv_vk * ziN;                 // :217
        const R ztmp2 = zsqrt_Cd * ziN;
        stab = nonneg(zeta_u);                                          // :220
        const R zChN = R(1.e-3) * zsqrt_CdN * (stab ? R(18.) : R(32.7));
        const R zCeN = R(1.e-3) * (R(34.6) * zsqrt_CdN);
        Ch = vmax(M::div(zChN * ztmp2, R(1.) + zChN * ztmp), K<R>::Cx_min);
        Ce = vmax(M::div(zCeN * ztmp2, R(1.) + zCeN * ztmp), K<R>::Cx_min);
        if (DIAG) { d_us = zus; d_1oL = z1oL; d_un10 = zUn10; d_chn = zChN; d_cen = zCeN; }
    }
    o.Cd = Cd; o.Ch = Ch; o.Ce = Ce; o.t_zu = t_zu; o.q_zu = q_zu; o.Ubzu = Ubzu; o.T_s = sst; o.q_s = ssq;
    if (DIAG) {   // optional outputs :229-235 ; z0_from_Cd without psi mod_phymbl.f90:1349
        o.CdN = zCdN; o.CeN = d_cen; o.ChN = d_chn; o.UN10 = d_un10; o.L = M::rcp(d_1oL); o.us = d_us;
        o.z0 = vmin(h.zu * M::exp(-M::div(vk, M::sqrt_pos(zCdN))), K<R>::z0_sea_max);
        o.dT_cs = R(0.); o.dT_wl = R(0.); o.Hz_wl = R(0.);
    }
}

// ---------------------------------------------------------------- ANDREAS (mod_blk_andreas.f90, Andreas et al. 2015)
// psi_m_andreas :307-360 — Paulson unstable, Grachev et al. (2007) stable
template <class R> __device__ __forceinline__ R psi_m_andreas(R pz)
{
    AB_REGION("psi_m_andreas");
    using M = Mth<R>;
    const R z = vmin(pz, R(15.));
    if (nonneg(z)) {
        const R zam = R(5.), zbm = R(5. / 6.5), zsr3 = R(1.7320508075688772);
        const R zbbm = R(0.6694329500821695);  // ((1-b_m)/b_m)^(1/3) = 0.3^(1/3)
        const R x = M::cbrt(M::abs(R(1.) + z));
        const R l1 = (x + zbbm) * R(1. / (1. + 0.6694329500821695));
        const R l2 = (x * x - x * zbbm + zbbm * zbbm) * R(1. / (1. - 0.6694329500821695 + 0.6694329500821695 * 0.6694329500821695));
        return R(-3. * 5. / (5. / 6.5)) * (x - R(1.))
               + zam * zbbm / (R(2.) * zbm)
                     * (M::log(M::div(l1 * l1, l2))
                        + R(2.) * zsr3 * (M::atan((R(2.) * x - zbbm) * R(1. / (1.7320508075688772 * 0.6694329500821695)))
                                          - R(0.8539936329836121)));  // ATAN((2-B_m)/(sqrt(3) B_m))
    }
    R m;
    psik<R>(M::abs(R(1.) - R(16.) * z), &m, nullptr);      // x2 = MAX(SQRT(ABS(1 - 16 zeta)), 1): the MAX never binds for zeta < 0
    return m;
}
// psi_h_andreas :363-410
template <class R> __device__ __forceinline__ R psi_h_andreas(R pz)
{
    AB_REGION("psi_h_andreas");
    using M = Mth<R>;
    const R z = vmin(pz, R(15.));
    if (nonneg(z)) {
        const R zah = R(5.), zbh = R(5.), zch = R(3.), zbbh = R(2.23606797749979);
        const R zz = R(2.) * z + zch;
        // LOG|(zz-B)/(zz+B)| - LOG|(c-B)/(c+B)|
        const R r = M::div(zz - zbbh, zz + zbbh) * R((3. + 2.23606797749979) / (3. - 2.23606797749979));
        return R(-0.5) * zbh * M::log(M::abs(R(1.) + zch * z + z * z))
               + (-zah / zbbh + R(0.5) * zbh * zch / zbbh) * M::log(M::abs(r));
    }
    const R x2 = vmax(M::sqrt_pos(M::abs(R(1.) - R(16.) * z)), R(1.));
    return R(2.) * M::log(R(0.5) * (R(1.) + x2));
}
// z0tq_LKB mod_phymbl.f90:1635-1701 (Liu, Katsaros & Businger 1979): both z0t (iflag 1) and z0q (iflag 2)
template <class R> __device__ __forceinline__ void z0tq_lkb(R zrr, R pz0, R &z0t, R &z0q)
{
    AB_REGION("z0tq_lkb");
    using M = Mth<R>;
    R rt = R(-999.), rq = R(-999.);
    if ((zrr > R(0.)) && (zrr < R(1000.))) {
        R at, bt, aq, bq;
        if (zrr <= R(0.11))       { at = R(0.177);   bt = R(0.);     aq = R(0.292);   bq = R(0.); }
        else if (zrr <= R(0.825)) { at = R(1.376);   bt = R(0.929);  aq = R(1.808);   bq = R(0.826); }
        else if (zrr <= R(3.0))   { at = R(1.026);   bt = R(-0.599); aq = R(1.393);   bq = R(-0.528); }
        else if (zrr <= R(10.0))  { at = R(1.625);   bt = R(-1.018); aq = R(1.956);   bq = R(-0.870); }
        else if (zrr <= R(30.0))  { at = R(4.661);   bt = R(-1.475); aq = R(4.994);   bq = R(-1.297); }
        else if (zrr <= R(100.))  { at = R(34.904);  bt = R(-2.067); aq = R(30.709);  bq = R(-1.845); }
        else if (zrr <= R(300.))  { at = R(1667.19); bt = R(-2.907); aq = R(1448.68); bq = R(-2.682); }
        else                      { at = R(5.88e5);  bt = R(-3.935); aq = R(2.98e5);  bq = R(-3.616); }
        const R lr = M::log(zrr);
        const R zs = M::div(pz0, zrr);
        rt = at * M::exp(bt * lr) * zs;
        rq = aq * M::exp(bq * lr) * zs;
    }
    z0t = vmin(vmax(M::abs(rt), R(1.E-9)), R(0.05));
    z0q = vmin(vmax(M::abs(rq), R(1.E-9)), R(0.05));
}
// turb_andreas :66-272
template <class R, bool DIAG = false, class A = R>
__device__ __forceinline__ void turb_andreas(const Heights<R> &h, const CellIn<R, A> &in, int nb_iter, CellOut<R, A> &o)
{
    using M = Mth<R>;
    const R vk = K<R>::vkarmn;
    const A psst = in.sst, pssq = in.ssq;
    const R pUbzu = vmax(R(0.25), in.wnd);                              // :157
    const R ziUb = M::rcp(pUbzu);
    R UN10 = pUbzu;
    A pt_zu = in.theta_zt, pq_zu = in.q_zt;
    R t_star = R(0.03316624790355400) * R(pt_zu - psst);                // Ch/SQRT(Cd), Cd=Ch=Ce=1.1e-3 :161-170
    R q_star = R(0.03316624790355400) * R(pq_zu - pssq);
    R RiB = ri_bulk<R, A>(h.zu, psst, pt_zu, pssq, pq_zu, pUbzu);       // :173
    R u_star = R(0.);
    R d_z0 = R(0.), d_zeta = R(0.);   // DIAG only
#pragma unroll 1
    for (int jit = 1; jit <= nb_iter; ++jit) {
        if (RiB < R(0.15)) {                                            // :183-191
            const R za = UN10 - R(8.271);                               // u_star_andreas_sclr :289-291
            u_star = R(0.239) + R(0.0433) * (za + M::sqrt_pos(R(0.12) * za * za + R(0.181)));
        } else {
            u_star = R(0.01) * pUbzu;                                   // SQRT(Cx_min) = 1e-2
        }
        const R zeta_u = h.zu * one_on_l(R(pt_zu), R(pq_zu), u_star, t_star, q_star);  // :200
        R ztmp0 = u_star * ziUb;
        const R pCd = vmax(ztmp0 * ztmp0, K<R>::Cx_min);                // :209
        const R psm = psi_m_andreas<R>(zeta_u);
        const R z0 = vmin(h.zu * M::exp(-(M::div(vk, M::sqrt_pos(pCd)) + psm)), K<R>::z0_sea_max);  // :214
        if (DIAG) { d_z0 = z0; d_zeta = zeta_u; }
        ztmp0 = M::div(z0 * u_star, visc_air(R(pt_zu)));                // :219 Re_r
        R z0t, z0q;
        z0tq_lkb(ztmp0, z0, z0t, z0q);                                  // :220-221
        const R psh = psi_h_andreas<R>(zeta_u);
        t_star = M::div(R(pt_zu - psst) * vk, h.log_zu - M::log(z0t) - psh);  // :226-227
        q_star = M::div(R(pq_zu - pssq) * vk, h.log_zu - M::log(z0q) - psh);
        if ((!h.zt_eq_zu) && (jit > 1)) {                               // :229-236
            const R zeta_t = zeta_u * h.zt_o_zu;
            const R zp = h.log_ztu + psh - psi_h_andreas<R>(zeta_t);
            pt_zu = in.theta_zt - A(t_star * K<R>::inv_vk * zp);
            pq_zu = in.q_zt - A(q_star * K<R>::inv_vk * zp);
            RiB = ri_bulk<R, A>(h.zu, psst, pt_zu, pssq, pq_zu, pUbzu);
        }
        UN10 = vmax(R(0.1), pUbzu - u_star * K<R>::inv_vk * (h.log_zu10 - psm));  // :239 (UN10_from_ustar mod_phymbl.f90:1508)
    }
    const R ztmp0 = u_star * ziUb;                                      // :247-254
    o.Cd = vmax(ztmp0 * ztmp0, K<R>::Cx_min);
    const R d1 = sfloor(R(pt_zu - psst), R(1.E-6));
    const R d2 = sfloor(R(pq_zu - pssq), R(1.E-9));
    o.Ch = vmax(M::div(ztmp0 * t_star, d1), R(0.35E-3));
    o.Ce = vmax(M::div(ztmp0 * q_star, d2), R(0.35E-3));
    o.t_zu = pt_zu; o.q_zu = pq_zu; o.Ubzu = pUbzu; o.T_s = psst; o.q_s = pssq;
    if (DIAG) {   // optional outputs :256-267
        const R zi = M::rcp(M::log(M::div(h.zu, d_z0)));
        o.CdN = vmax(K<R>::vkarmn2 * zi * zi, K<R>::Cx_min);
        R z0t, z0q;
        z0tq_lkb(M::div(d_z0 * u_star, visc_air(R(pt_zu))), d_z0, z0t, z0q);
        o.ChN = M::div(K<R>::vkarmn2 * zi, M::log(M::div(h.zu, z0t)));
        o.CeN = M::div(K<R>::vkarmn2 * zi, M::log(M::div(h.zu, z0q)));
        o.z0 = d_z0; o.us = u_star; o.L = M::div(h.zu, d_zeta);
        o.UN10 = pUbzu - u_star * K<R>::inv_vk * (h.log_zu10 - psi_m_andreas<R>(d_zeta));
        o.dT_cs = R(0.); o.dT_wl = R(0.); o.Hz_wl = R(0.);
    }
}

}  // namespace ab
