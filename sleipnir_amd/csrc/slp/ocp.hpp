// slp::OCP — the reference's optimal-control convenience layer over Problem
// (include/sleipnir/optimization/ocp.hpp:49-400, ocp/{dynamics_type,timestep_method,
// transcription_method}.hpp): it owns X (states x samples), U (inputs x samples) and the
// per-sample timesteps, and turns a dynamics function into the constraints of a direct
// transcription, a Hermite-Simpson direct collocation or a single-shooting rollout.
// Host-side model building only: what it produces is an ordinary Problem whose solve() runs
// the GPU path.  Same names, argument meaning and defaults as the reference so that its OCP
// programs (test/src/optimization/*_ocp_test.cpp) compile against this header.
#pragma once

#include <chrono>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <utility>

#include "problem.hpp"

namespace slp {

// ocp/dynamics_type.hpp:9-14
enum class DynamicsType : uint8_t {
  EXPLICIT_ODE,  // dx/dt = f(t, x, u)
  DISCRETE,      // x_{k+1} = f(t, x_k, u_k, dt)
};

// ocp/timestep_method.hpp:9-17
enum class TimestepMethod : uint8_t {
  FIXED,            // a constant
  VARIABLE,         // one decision variable per sample
  VARIABLE_SINGLE,  // one decision variable shared by all samples
};

// ocp/transcription_method.hpp:9-19
enum class TranscriptionMethod : uint8_t {
  DIRECT_TRANSCRIPTION,  // states are decision variables tied together by equality constraints
  DIRECT_COLLOCATION,    // cubic Hermite splines between samples, dynamics enforced at the midpoints
  SINGLE_SHOOTING,       // states are expressions of the inputs and the initial state
};

template <typename Scalar>
class OCP;

template <>
class OCP<double> : public Problem<double> {
 public:
  using Matrix = VariableMatrix<double>;
  using Var = Variable<double>;
  using TimedDynamics = std::function<Matrix(const Var& t, const Matrix& x, const Matrix& u, const Var& dt)>;
  using Dynamics = std::function<Matrix(const Matrix& x, const Matrix& u)>;

  // ocp.hpp:68-90: dynamics without explicit time dependence
  OCP(int num_states, int num_inputs, std::chrono::duration<double> dt, int num_steps, Dynamics dynamics,
      DynamicsType dynamics_type = DynamicsType::EXPLICIT_ODE, TimestepMethod timestep_method = TimestepMethod::FIXED,
      TranscriptionMethod transcription_method = TranscriptionMethod::DIRECT_TRANSCRIPTION)
      : OCP{num_states,
            num_inputs,
            dt,
            num_steps,
            TimedDynamics{[f = std::move(dynamics)](const Var&, const Matrix& x, const Matrix& u, const Var&) {
              return f(x, u);
            }},
            dynamics_type,
            timestep_method,
            transcription_method} {}

  // ocp.hpp:107-157
  OCP(int num_states, int num_inputs, std::chrono::duration<double> dt, int num_steps, TimedDynamics dynamics,
      DynamicsType dynamics_type = DynamicsType::EXPLICIT_ODE, TimestepMethod timestep_method = TimestepMethod::FIXED,
      TranscriptionMethod transcription_method = TranscriptionMethod::DIRECT_TRANSCRIPTION)
      : m_samples{num_steps + 1}, m_f{std::move(dynamics)}, m_kind{dynamics_type} {
    // one input more than there are steps: the last sample's constraint functions need one too (:122-123)
    m_U = this->decision_variable(num_inputs, m_samples);
    m_DT = Matrix{1, m_samples};
    switch (timestep_method) {
      case TimestepMethod::FIXED:
        for (int k = 0; k < m_samples; ++k) m_DT[0, k] = dt.count();
        break;
      case TimestepMethod::VARIABLE_SINGLE: {
        Var shared = this->decision_variable();
        shared.set_value(dt.count());
        for (int k = 0; k < m_samples; ++k) m_DT[0, k] = shared;
        break;
      }
      case TimestepMethod::VARIABLE:
        m_DT = this->decision_variable(1, m_samples);
        for (int k = 0; k < m_samples; ++k) m_DT[0, k].set_value(dt.count());
        break;
    }
    if (transcription_method == TranscriptionMethod::SINGLE_SHOOTING) {
      m_X = Matrix{num_states, m_samples};  // expressions, filled by the rollout
    } else {
      m_X = this->decision_variable(num_states, m_samples);
    }
    if (transcription_method == TranscriptionMethod::DIRECT_COLLOCATION && m_kind != DynamicsType::EXPLICIT_ODE)
      throw std::invalid_argument("slp::OCP: direct collocation needs an explicit ODE");  // slp_assert at :323

    // One pass over the steps for all three methods: the running time is the sum of the
    // timesteps so far (a Variable: the timesteps may be decision variables).
    Var time{0.0};
    for (int k = 0; k + 1 < m_samples; ++k) {
      const Var h = m_DT[0, k];
      Matrix x0 = m_X.col(k);
      Matrix u0 = m_U.col(k);
      switch (transcription_method) {
        case TranscriptionMethod::DIRECT_TRANSCRIPTION:
          this->subject_to(Matrix{m_X.col(k + 1)} == step(time, x0, u0, h));  // :359-379
          break;
        case TranscriptionMethod::SINGLE_SHOOTING:
          m_X.col(k + 1) = step(time, x0, u0, h);  // :382-401
          break;
        case TranscriptionMethod::DIRECT_COLLOCATION: {  // :322-357 (Hermite-Simpson)
          Matrix x1 = m_X.col(k + 1);
          Matrix u1 = m_U.col(k + 1);
          const Var t1 = time + h;
          const Matrix f0 = m_f(time, x0, u0, h);
          const Matrix f1 = m_f(t1, x1, u1, h);
          const Matrix xdot_mid = -3.0 / (2.0 * h) * (x0 - x1) - 0.25 * (f0 + f1);
          const Var t_mid = time + 0.5 * h;
          const Matrix x_mid = 0.5 * (x0 + x1) + h / 8.0 * (f0 - f1);
          const Matrix u_mid = 0.5 * (u0 + u1);
          this->subject_to(xdot_mid == m_f(t_mid, x_mid, u_mid, h));
          break;
        }
      }
      time += h;
    }
  }

  // ocp.hpp:162-176
  template <typename T>
  void constrain_initial_state(const T& initial_state) {
    this->subject_to(this->initial_state() == initial_state);
  }
  template <typename T>
  void constrain_final_state(const T& final_state) {
    this->subject_to(this->final_state() == final_state);
  }

  // ocp.hpp:183-213: the callback sees every sample, the last one included
  void for_each_step(const std::function<void(const Matrix& x, const Matrix& u)>& callback) {
    for (int k = 0; k < m_samples; ++k) callback(Matrix{m_X.col(k)}, Matrix{m_U.col(k)});
  }
  void for_each_step(const std::function<void(const Var& t, const Matrix& x, const Matrix& u, const Var& dt)>& callback) {
    Var time{0.0};
    for (int k = 0; k < m_samples; ++k) {
      const Var h = m_DT[0, k];
      callback(time, Matrix{m_X.col(k)}, Matrix{m_U.col(k)}, h);
      time += h;
    }
  }

  // ocp.hpp:220-239
  template <typename T>
  void set_lower_input_bound(const T& lower_bound) {
    for (int k = 0; k < m_samples; ++k) this->subject_to(Matrix{m_U.col(k)} >= lower_bound);
  }
  template <typename T>
  void set_upper_input_bound(const T& upper_bound) {
    for (int k = 0; k < m_samples; ++k) this->subject_to(Matrix{m_U.col(k)} <= upper_bound);
  }

  // ocp.hpp:243-252
  void set_min_timestep(std::chrono::duration<double> min_timestep) { this->subject_to(m_DT >= min_timestep.count()); }
  void set_max_timestep(std::chrono::duration<double> max_timestep) { this->subject_to(m_DT <= max_timestep.count()); }

  // ocp.hpp:260-288
  Matrix& X() { return m_X; }
  Matrix& U() { return m_U; }
  Matrix& dt() { return m_DT; }
  Matrix initial_state() { return m_X.col(0); }
  Matrix final_state() { return m_X.col(m_samples - 1); }

 private:
  int m_samples;  // num_steps + 1
  TimedDynamics m_f;
  DynamicsType m_kind;
  Matrix m_X, m_U, m_DT;

  // x_{k+1} as an expression of (t, x_k, u_k, h): the state transition function itself, or one
  // classical Runge-Kutta step of the ODE (ocp.hpp:310-319)
  Matrix step(const Var& t, const Matrix& x, const Matrix& u, const Var& h) {
    if (m_kind == DynamicsType::DISCRETE) return m_f(t, x, u, h);
    const Var half = h * 0.5;
    const Matrix k1 = m_f(t, x, u, h);
    const Matrix k2 = m_f(t + half, x + k1 * half, u, h);
    const Matrix k3 = m_f(t + half, x + k2 * half, u, h);
    const Matrix k4 = m_f(t + h, x + k3 * h, u, h);
    return x + (k1 + k2 * 2.0 + k3 * 2.0 + k4) * (h / 6.0);
  }
};

}  // namespace slp
